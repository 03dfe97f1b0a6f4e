"""Host time stabilize_resident() takes to ISSUE one config-2 clip (no synchronisation) against the GPU's time per clip: one call alone, and 50 back to
back (the deferred status check of clip i - 2 bounds the run-ahead to two clips, so the issue time then follows the GPU).  MI355X box: 0.34 ms
alone, 1.22 ms per call in the pipeline at 1.27 ms per clip.     python tools/issue_time.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
F, H, W, R, C = 300, 1080, 1920, 16, 16
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100, device='cuda:0')
d_frames = synthetic.frames_torch(F, H, W, dev, seed=0)
d_disp = torch.from_numpy(disp).to(dev)
out = torch.empty_like(d_frames)
for _ in range(5):
    s.stabilize_resident(d_frames, d_disp, hom, out=out, check='deferred')
torch.cuda.synchronize()
for K in (1, 50):
    t0 = time.perf_counter()
    for _ in range(K):
        s.stabilize_resident(d_frames, d_disp, hom, out=out, check='deferred')
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'K={K}: host issue {1e3 * (t1 - t0) / K:.3f} ms per call, wall {1e3 * (t2 - t0) / K:.3f} ms per call')
s.finish()

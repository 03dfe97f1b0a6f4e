"""Host RSS, device memory and thread count over 200 host-to-host clips (stabilize_clip, crop=True) and 2,000 resident clips (stabilize_resident) of
120 x 1280x720 frames: flat from the first hundred on (MI355X box, end of round 5: 3,330 MiB / 2,394 MiB / 66 threads at every snapshot).
    python tools/leak_check.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, psutil
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
F, H, W, R, C = 120, 720, 1280, 16, 16
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
base = synthetic.frames_torch(8, H, W, dev, seed=0).cpu().numpy()
frames = [np.ascontiguousarray(base[i % 8]) for i in range(F)]
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100, device='cuda:0')
proc = psutil.Process()
def snap(tag):
    free, total = torch.cuda.mem_get_info(dev)
    print(f'{tag}: host RSS {proc.memory_info().rss / 2**20:.0f} MiB, device used {(total - free) / 2**20:.0f} MiB, threads {proc.num_threads()}', flush=True)
d_frames = torch.from_numpy(np.stack(frames)).to(dev)
d_disp = torch.from_numpy(disp).to(dev)
for rnd in range(4):
    for _ in range(50):
        r = s.stabilize_clip(frames, disp, hom, crop=True)
        del r
    for _ in range(500):
        s.stabilize_resident(d_frames, d_disp, hom, check='deferred')
    s.finish(); torch.cuda.synchronize()
    snap(f'after {50 * (rnd + 1)} host clips + {500 * (rnd + 1)} resident clips')

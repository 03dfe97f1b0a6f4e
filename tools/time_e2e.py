"""End-to-end stabilize_clip() with host (NumPy) buffers in and out: the PCIe-inclusive rate of the drop-in methods."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
H, W, F, R, C = 1080, 1920, 300, 16, 16
disp, hom = synthetic.motion(F, R, C, seed=0)
frames = synthetic.frames_torch(F, H, W, torch.device('cuda:0'), seed=0).cpu().numpy()
frame_list = list(frames)
s = MeshFlowStabilizer(device='cuda:0')
for mode in ('list', 'array'):
    inp = frame_list if mode == 'list' else frames
    for rep in range(3):
        t0 = time.perf_counter()
        out, bounds, stab, score = s.stabilize_clip(inp, disp, hom)
        dt = time.perf_counter() - t0
        print(f'{mode}: stabilize_clip {dt*1e3:.1f} ms -> {F/dt:.0f} frames/s   bounds={tuple(int(b) for b in bounds)} score={score:.4f}')
    t0 = time.perf_counter()
    out5 = s.stabilize_clip(inp, disp, hom, crop=True, keep_uncropped=False)
    dt = time.perf_counter() - t0
    print(f'{mode}: with crop+resize {dt*1e3:.1f} ms -> {F/dt:.0f} frames/s')

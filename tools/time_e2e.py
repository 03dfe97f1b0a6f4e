"""End-to-end stabilize_clip() with host (NumPy) buffers in and out: the PCIe-inclusive rate of the drop-in methods.
usage: time_e2e.py [chunk_frames io_threads]...   (pairs to sweep; default 16 3)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
H, W, F, R, C = 1080, 1920, 300, 16, 16
disp, hom = synthetic.motion(F, R, C, seed=0)
frames = synthetic.frames_torch(F, H, W, torch.device('cuda:0'), seed=0).cpu().numpy()
frame_list = [f.copy() for f in frames]          # separate allocations, like a decoder's output
s = MeshFlowStabilizer(device='cuda:0')
args = [int(v) for v in sys.argv[1:]] or [16, 3]
for chunk, threads in zip(args[0::2], args[1::2]):
    for mode in ('list', 'array'):
        inp = frame_list if mode == 'list' else frames
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            out, bounds, stab, score = s.stabilize_clip(inp, disp, hom, chunk_frames=chunk, io_threads=threads)
            best = min(best, time.perf_counter() - t0)
            del out
        t0 = time.perf_counter()
        out5 = s.stabilize_clip(inp, disp, hom, crop=True, keep_uncropped=False, chunk_frames=chunk, io_threads=threads)
        dt = time.perf_counter() - t0
        del out5
        print(f'chunk={chunk} threads={threads} {mode}: stabilize_clip best {best*1e3:.1f} ms -> {F/best:.0f} frames/s; '
              f'crop+resize, cropped only: {dt*1e3:.1f} ms -> {F/dt:.0f} frames/s   bounds={tuple(int(b) for b in bounds)}', flush=True)

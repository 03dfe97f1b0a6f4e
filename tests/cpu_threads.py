import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from meshflow_amd import synthetic
from oracle import clib, meshflow_oracle as mo
H, W, F, R, C = 1080, 1920, 300, 16, 16
disp, hom = synthetic.motion(F, R, C, seed=0)
taps, lam, on = mo.jacobi_band_coefficients(F, W, H, 0, hom, 10)
b = np.ascontiguousarray(disp.reshape(F, -1))
frames = np.ascontiguousarray(np.broadcast_to(synthetic.frames_numpy(1, H, W, seed=0), (128, H, W, 3)))
print('cpu_count', os.cpu_count())
for th in (16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    clib.set_threads(th)
    t0 = time.perf_counter(); stab = clib.jacobi_banded(b, taps, lam, np.reciprocal(on), 10, 100, openmp=True); tj = time.perf_counter() - t0
    st = stab.reshape(disp.shape)
    t0 = time.perf_counter(); clib.warp_clip(frames, R, C, disp[:128], st[:128], use_bbox=True, openmp=True); tw = time.perf_counter() - t0
    print(f'threads={th}: jacobi {tj:.3f} s, warp 128 frames {tw:.3f} s -> {128/tw:.1f} frames/s')

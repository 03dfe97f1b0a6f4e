"""frames/s of mf_warp_u8c3_host (raw ctypes, no torch): cfg2-sized clip, pageable NumPy buffers and pinned buffers."""
import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from meshflow_amd import _lib, synthetic
F, H, W, R, C = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 1080, 1920, 16, 16
frames = np.ascontiguousarray(np.broadcast_to(synthetic.frames_numpy(4, H, W, seed=0), (F // 4, 4, H, W, 3)).reshape(F, H, W, 3))
disp, hom = synthetic.motion(F, R, C, seed=0)
stab = np.ascontiguousarray(0.3 * disp)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
out = np.empty_like(frames); crop = np.zeros((F, 4), np.int32); ms = ctypes.c_float(0)
border = (ctypes.c_uint8 * 3)(0, 0, 255)
for label in ('pageable', 'pinned'):
    if label == 'pinned':
        hin, hout = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(_lib.lib.mf_malloc_host(ctypes.byref(hin), frames.nbytes)); _lib.check(_lib.lib.mf_malloc_host(ctypes.byref(hout), frames.nbytes))
        ctypes.memmove(hin, frames.ctypes.data, frames.nbytes)
        a, b = hin, hout
    else:
        a, b = p(frames), p(out)
    t = []
    for i in range(7):
        t0 = time.perf_counter()
        _lib.check(_lib.lib.mf_warp_u8c3_host(a, b, p(disp), p(stab), F, W, H, R, C, border, p(crop), ctypes.byref(ms)))
        t.append(time.perf_counter() - t0)
    t = t[1:]
    print(f'{label}: mean {F / np.mean(t):.0f} frames/s, best {F / np.min(t):.0f} frames/s ({np.mean(t) * 1e3:.1f} ms per clip, kernels {ms.value:.2f} ms)')

"""Why does a freshly allocated output buffer cost +150 ms?  Times page population (1 / 4 / 8 threads, madvise vs touch) and the
host pipeline on populated-but-never-used vs reused buffers."""
import ctypes, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
libc = ctypes.CDLL(None, use_errno=True)
N = 1866240000


def populate(a, nthreads, how):
    lo = (a.ctypes.data + 4095) & ~4095
    hi = (a.ctypes.data + a.nbytes) & ~4095
    step = ((hi - lo) // nthreads + 4095) & ~4095
    def work(t):
        b, e = lo + t * step, min(lo + (t + 1) * step, hi)
        if e <= b: return
        if how == 'madvise':
            r = libc.madvise(ctypes.c_void_p(b), ctypes.c_size_t(e - b), 23)
            if r != 0: print('madvise failed errno', ctypes.get_errno())
        else:
            v = np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)), shape=(e - b,))
            v[::4096] = 0
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0


for how in ('madvise', 'touch'):
    for nt in (1, 4, 8):
        a = np.empty(N, dtype=np.uint8)
        print(f'populate 1.87 GB {how} x{nt}: {populate(a, nt, how) * 1e3:.1f} ms', flush=True)
        del a

from meshflow_amd import _lib, synthetic
F, H, W, R, C = 300, 1080, 1920, 16, 16
frames = np.ascontiguousarray(np.broadcast_to(synthetic.frames_numpy(4, H, W, seed=0), (F // 4, 4, H, W, 3)).reshape(F, H, W, 3))
disp, hom = synthetic.motion(F, R, C, seed=0)
stab = np.ascontiguousarray(0.3 * disp)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
border = (ctypes.c_uint8 * 3)(0, 0, 255)
crop = np.zeros((F, 4), np.int32)


def call(out):
    t0 = time.perf_counter()
    _lib.check(_lib.lib.mf_warp_u8c3_host(p(frames), p(out), p(disp), p(stab), F, W, H, R, C, border, p(crop), None))
    return (time.perf_counter() - t0) * 1e3


out = np.empty_like(frames)
print('first call (fresh out): %.1f ms' % call(out))
print('second call (same out): %.1f ms' % call(out))
print('third call (same out): %.1f ms' % call(out))
for k in range(3):
    o = np.empty_like(frames)
    tp = populate(o.reshape(-1), 8, 'touch') * 1e3
    print('fresh out, populated by 8 touch threads first (%.1f ms): call %.1f ms' % (tp, call(o)))
    del o
os.environ['X'] = '1'
for k in range(2):
    o = np.empty_like(frames)
    print('fresh out, unpopulated: call %.1f ms' % call(o))
    del o

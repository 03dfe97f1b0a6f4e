"""How fast is in-place page-locking (hipHostRegister) of a NumPy clip, and the H2D / D2H it enables?
Every repetition uses arrays that were never registered before (the one-shot situation of stabilize_clip)."""
import time, numpy as np, torch
n = 1920 * 1080 * 3 * 300
rt = torch.cuda.cudart()
d = torch.empty(n, dtype=torch.uint8, device='cuda')
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
for rep in range(3):
    a = src.copy()                                   # fresh, touched pages
    t0 = time.perf_counter(); rt.cudaHostRegister(a.ctypes.data, n, 0); t1 = time.perf_counter()
    d.copy_(torch.from_numpy(a), non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    rt.cudaHostUnregister(a.ctypes.data); t3 = time.perf_counter()
    print(f'in : register {1e3*(t1-t0):.1f} ms  H2D {1e3*(t2-t1):.1f} ms  unregister {1e3*(t3-t2):.1f} ms  total {1e3*(t3-t0):.1f} ms ({n/(t3-t0)/1e9:.1f} GB/s)')
    out = np.empty(n, dtype=np.uint8)                # fresh, untouched pages
    t0 = time.perf_counter(); rt.cudaHostRegister(out.ctypes.data, n, 0); t1 = time.perf_counter()
    torch.from_numpy(out).copy_(d, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    rt.cudaHostUnregister(out.ctypes.data); t3 = time.perf_counter()
    print(f'out: register {1e3*(t1-t0):.1f} ms  D2H {1e3*(t2-t1):.1f} ms  unregister {1e3*(t3-t2):.1f} ms  total {1e3*(t3-t0):.1f} ms ({n/(t3-t0)/1e9:.1f} GB/s)  ok={bool((out[:64] == a[:64]).all())}')
    del a, out
a = src.copy(); out = np.empty(n, dtype=np.uint8)
t0 = time.perf_counter(); d.copy_(torch.from_numpy(a)); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'pageable H2D {1e3*(t1-t0):.1f} ms ({n/(t1-t0)/1e9:.1f} GB/s)')
t0 = time.perf_counter(); torch.from_numpy(out).copy_(d); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'pageable D2H into untouched pages {1e3*(t1-t0):.1f} ms ({n/(t1-t0)/1e9:.1f} GB/s)')

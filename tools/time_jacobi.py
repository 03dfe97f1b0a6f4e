import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic, host
dev = torch.device('cuda:0')
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for F, R, C, om, it in ((600,32,32,10,100),(1200,32,32,10,100),(2400,32,32,10,100),(300,16,16,10,100),(600,16,16,10,100),(1200,16,16,10,100),(2400,16,16,10,100),(600,32,32,30,200),(1200,32,32,30,200)):
    disp, hom = synthetic.motion(F, R, C, seed=0)
    taps, lam, inv_on = host.jacobi_band_coefficients(F, 1920, 1080, 0, hom, om)
    b = torch.from_numpy(disp.reshape(F, -1)).to(dev)
    tt = [torch.from_numpy(a).to(dev) for a in (taps, lam, inv_on)]
    out = torch.empty_like(b)
    ms = t(lambda: ops.jacobi(b, *tt, om, it, out=out))
    flops = it * F * b.shape[1] * (2 * (2 * om + 1) + 3)
    print(f'F={F} mesh={R}x{C} omega={om} iters={it}: {ms:.3f} ms  {flops/ms/1e9:.2f} TFLOP/s')
import time
for F in (300, 2400):
    disp, hom = synthetic.motion(F, 16, 16, seed=0)
    t0 = time.perf_counter()
    for _ in range(20):
        host.jacobi_band_coefficients(F, 1920, 1080, 0, hom, 10)
    print(f'host coefficient set-up F={F}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms')

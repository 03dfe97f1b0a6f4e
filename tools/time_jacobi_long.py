import sys, os, numpy as np, torch, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from meshflow_amd import ops, synthetic
from oracle import clib
dev = torch.device('cuda:0')
for F, omega, iters in ((20000, 30, 20), (12000, 30, 20), (12000, 30, 200), (9000, 30, 20)):
    S = 578
    b = np.cumsum(2.0 * synthetic.normal(np.arange(F * S).reshape(F, S), seed=F + omega), axis=0)
    taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
    lam = 0.95 * synthetic.uniform01(np.arange(F), seed=3)
    inv_on = 1.0 / (1 + 2 * lam * taps.sum())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = (t(b), t(taps), t(lam), t(inv_on), omega, iters)
    x = ops.jacobi(*args); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.jacobi(*args, out=x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True) if iters <= 20 else None
    ok = None if want is None else bool(np.array_equal(x.cpu().numpy(), want))
    print(f'TILE30={os.environ.get("MF_JACOBI_TILE30")} F={F} omega={omega} iters={iters}: {ms:.3f} ms, {iters*F*S*(2*(2*omega+1)+3)/ms/1e9:.1f} TFLOP/s, exact={ok}', flush=True)

"""Interleaved A/B of several builds of libmeshflow_hip.so on the Jacobi sweep (raw C ABI), alternating order within a round.

    python tools/ab_jacobi.py [cfg2|cfg3] libA.so libB.so ...       (the first build's output is the reference: max |diff| is printed)
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic, host
wl = sys.argv[1]
F, R, C, om, it = {'cfg2': (300, 16, 16, 10, 100), 'cfg3': (600, 32, 32, 30, 200)}[wl] if wl in ('cfg2', 'cfg3') else tuple(int(v) for v in wl.split(','))   # or F,R,C,omega,iters
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
taps, lam, inv_on = host.jacobi_band_coefficients(F, 1920, 1080, 0, hom, om)
b = torch.from_numpy(disp.reshape(F, -1)).to(dev)
S = b.shape[1]
tt = [torch.from_numpy(a).to(dev) for a in (taps, lam, inv_on)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
vp, i = ctypes.c_void_p, ctypes.c_int
libs = []
for path in sys.argv[2:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.mf_jacobi_f64.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    libs.append((os.path.basename(path), lib, torch.empty_like(b), []))
def run(v):
    assert v[1].mf_jacobi_f64(b.data_ptr(), v[2].data_ptr(), tt[0].data_ptr(), tt[1].data_ptr(), tt[2].data_ptr(), F, S, om, it, st) == 0
for v in libs: run(v); run(v)
torch.cuda.synchronize()
for v in libs[1:]:
    print(f'{wl}: {v[0]} vs {libs[0][0]}: max |diff| {float((v[2] - libs[0][2]).abs().max()):.3e}')
for rnd in range(8):
    for v in (libs if rnd % 2 == 0 else libs[::-1]):
        run(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(v)
        e1.record(); torch.cuda.synchronize()
        v[3].append(e0.elapsed_time(e1) / 10)
for v in libs:
    print(f'{wl} {v[0]:24s} sweep median {np.median(v[3]) * 1e3:7.1f} us  min {min(v[3]) * 1e3:7.1f}')

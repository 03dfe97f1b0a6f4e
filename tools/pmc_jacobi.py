"""One Jacobi workload, a few launches (for the counter passes of tools/pmc_kernel.sh):  python tools/pmc_jacobi.py [cfg2|cfg3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meshflow_amd import ops, synthetic, host
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
F, R, C, om, it = {'cfg2': (300, 16, 16, 10, 100), 'cfg3': (600, 32, 32, 30, 200)}[wl]
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
taps, lam, inv_on = host.jacobi_band_coefficients(F, 1920, 1080, 0, hom, om)
b = torch.from_numpy(disp.reshape(F, -1)).to(dev)
tt = [torch.from_numpy(a).to(dev) for a in (taps, lam, inv_on)]
out = torch.empty_like(b)
for _ in range(3):
    ops.jacobi(b, *tt, om, it, out=out)
torch.cuda.synchronize()
print('done', wl, b.shape)

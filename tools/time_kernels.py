"""Per-kernel HIP-event timings of the cfg2 step (cell table incl. plan, warp, Jacobi)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
H, W, F, R, C = 1080, 1920, 300, 16, 16
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(device='cuda:0')
d_disp = torch.from_numpy(disp).to(dev)
d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
frames = synthetic.frames_torch(F, H, W, dev, seed=0)
out = torch.empty_like(frames)
table = ops.CellTable(F, W, H, R, C, dev)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('cell_table+plan ms', t(lambda: ops.cell_table(d_disp, d_stab, W, H, R, C, table=table)))
print('warp ms', t(lambda: ops.warp(frames, table, out=out)))
print('jacobi ms (incl host coeffs)', t(lambda: s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)))
bounds = (13, 11, 1909, 1068)
out2 = torch.empty_like(frames)
print('crop_resize ms', t(lambda: ops.crop_resize(out, bounds, out=out2)))

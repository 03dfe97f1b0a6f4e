"""Warp with a 1x1 mesh (every interior footprint has exactly one IN cell): cost of the single-owner path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
H, W, F, R, C = 1080, 1920, 300, 1, 1
dev = torch.device('cuda:0')
z = np.zeros((F, R + 1, C + 1, 2))
s = z.copy(); s[..., 0] = 3.3; s[..., 1] = -2.7; s[:, 1, 1, 0] += 2.0      # a shift plus a little perspective
frames = synthetic.frames_torch(F, H, W, dev, seed=0)
out = torch.empty_like(frames)
table = ops.cell_table(torch.from_numpy(z).to(dev), torch.from_numpy(s).to(dev), W, H, R, C)
for _ in range(3): ops.warp(frames, table, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.warp(frames, table, out=out)
e1.record(); torch.cuda.synchronize()
print('1x1 mesh warp ms', e0.elapsed_time(e1) / 5)

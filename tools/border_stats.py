"""Shapes of the footprints that take the warp kernel's general path (not HOT / PAIR / MULTI): where they sit and what their lists look like."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
H, W, F, R, C, omega, iters = 1080, 1920, 300, 16, 16, 10, 100
if len(sys.argv) > 1 and sys.argv[1] == 'cfg3':
    F, R, C, omega, iters = 600, 32, 32, 30, 200
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters)
d_disp = torch.from_numpy(disp).to(dev)
d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
n = 100
sel = slice(100, 100 + n)
table = ops.cell_table(d_disp[sel], d_stab[sel], W, H, R, C)
torch.cuda.synchronize()
buf = table.buf.cpu().numpy()
nrec = n * R * C
plan_off = (nrec * (32 * 8 + 8 + 28 * 4) + 15) & ~15
nfx, nfy = (W + 31) // 32, (H + 7) // 8
npl = n * nfx * nfy
plan = buf[plan_off:plan_off + npl * 16].view(np.uint16).reshape(n, nfy, nfx, 8)
regs = buf[plan_off + npl * 16: plan_off + npl * 24].view(np.uint32)[0::2].reshape(n, nfy, nfx)
hot = (plan[..., 1] & 0x6000) == 0x2000
valid = (plan & 0x4000) != 0
overflow = plan[..., 7] == 0xFFFF
ne = np.where(overflow, 9, valid[..., :4].sum(-1) + np.where(valid[..., :4].all(-1), valid[..., 4:].sum(-1), 0))
pair = (ne == 2) & ((plan[..., 2] & 0x6000) == 0x2000)
multi = (ne >= 2) & (ne <= 4) & ((plan[..., 4] & 0x2000) != 0) & ~pair
general = ~hot & ~pair & ~multi
staged = (regs >> 31) & 1; deep = (regs >> 30) & 1; noflag = (regs >> 28) & 1
yy, xx = np.meshgrid(np.arange(nfy), np.arange(nfx), indexing='ij')
edge = (yy == 0) | (yy == nfy - 1) | (xx == 0) | (xx == nfx - 1)
ring2 = ((yy <= 1) | (yy >= nfy - 2) | (xx <= 0) | (xx >= nfx - 1))
print(f'footprints {npl}: hot {hot.mean():.4f} pair {pair.mean():.4f} multi {multi.mean():.4f} general {general.mean():.4f}')
g = general
print(f'general: on the outermost footprint ring {(g & edge[None]).sum() / g.sum():.3f}; rows 0-1 / last two or first / last column {(g & ring2[None]).sum() / g.sum():.3f}')
print(f'general by candidates:', {int(k): round(float(((ne == k) & g).sum() / g.sum()), 4) for k in np.unique(ne[g])})
inn0 = (plan[..., 0] & 0xC000) == 0xC000
print(f'general, one IN cell: {(g & (ne == 1) & inn0).sum() / g.sum():.3f}; staged {(g & (staged == 1)).sum() / g.sum():.3f}; staged & not deep {(g & (staged == 1) & (deep == 0)).sum() / g.sum():.3f}; noflag {(g & (noflag == 1)).sum() / g.sum():.3f}')
print(f'needs scan (not deep, not noflag): {((deep == 0) & (noflag == 0)).mean():.4f}')
top = g & (yy == 0)[None]; bot = g & (yy == nfy - 1)[None]; lef = g & (xx == 0)[None]; rig = g & (xx == nfx - 1)[None]
print(f'general on top row {top.sum() / g.sum():.3f} bottom row {bot.sum() / g.sum():.3f} left col {lef.sum() / g.sum():.3f} right col {rig.sum() / g.sum():.3f}; elsewhere {(g & ~edge[None]).sum() / g.sum():.3f}')
inner = g & ~edge[None]
print('general away from the ring by candidates:', {int(k): round(float(((ne == k) & inner).sum() / max(inner.sum(), 1)), 4) for k in np.unique(ne[inner])} if inner.sum() else {})

"""Source-window extents of the hot footprints (one IN cell, certified): how many rows / bytes a tight window needs.

    python tools/window_stats.py [cfg2|cfg3|cfg4shard]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
WORK = {'cfg2': (1080, 1920, 300, 16, 16, 10, 100), 'cfg3': (1080, 1920, 600, 32, 32, 30, 200), 'cfg4shard': (2160, 3840, 150, 16, 16, 10, 100)}
wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
H, W, F, R, C, omega, iters = WORK[wl]
SLACK = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0 / 16
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters)
d_disp = torch.from_numpy(disp).to(dev)
d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
n = 40
sel = slice(F // 2, F // 2 + n)
table = ops.cell_table(d_disp[sel], d_stab[sel], W, H, R, C)
torch.cuda.synchronize()
buf = table.buf.cpu().numpy()
nrec = n * R * C
rec = buf[:nrec * 32 * 8].view(np.float64).reshape(n, R * C, 32)
plan_off = (nrec * (32 * 8 + 8 + 28 * 4) + 15) & ~15
nfx, nfy = (W + 31) // 32, (H + 7) // 8
npl = n * nfx * nfy
plan = buf[plan_off:plan_off + npl * 16].view(np.uint16).reshape(n, nfy, nfx, 8)
hot = (plan[..., 1] & 0x6000) == 0x2000
print(f'{wl}: hot {hot.mean():.4f} of {npl} footprints')
f_i, y_i, x_i = np.nonzero(hot)
k = (plan[..., 0] & 0xFFF)[hot]
Hi = rec[f_i, k, 9:18]
xa, ya = x_i * 32.0, y_i * 8.0
us, vs = [], []
for cx, cy in ((xa, ya), (xa + 31, ya), (xa, ya + 7), (xa + 31, ya + 7)):
    w = Hi[:, 6] * cx + Hi[:, 7] * cy + Hi[:, 8]
    us.append((Hi[:, 0] * cx + Hi[:, 1] * cy + Hi[:, 2]) / w)
    vs.append((Hi[:, 3] * cx + Hi[:, 4] * cy + Hi[:, 5]) / w)
us, vs = np.array(us), np.array(vs)
ix_lo = np.floor(us.min(0) - SLACK); ix_hi = np.floor(us.max(0) + SLACK) + 1
iy_lo = np.floor(vs.min(0) - SLACK); iy_hi = np.floor(vs.max(0) + SLACK) + 1
rows = (iy_hi - iy_lo + 1).astype(int)
bs = (3 * ix_lo.astype(int)) & ~3
nbytes = (3 * ix_hi.astype(int) + 3 - bs)
print('rows histogram:', {int(r): round(float((rows == r).mean()), 4) for r in np.unique(rows)})
print('bytes histogram (16-B chunks):', {int(c): round(float((np.ceil(nbytes / 16) == c).mean()), 4) for c in np.unique(np.ceil(nbytes / 16))})
for rr, cc in ((9, 7), (10, 6), (8, 8), (9, 112 // 16), (10, 7), (12, 8), (16, 8), (11, 7)):
    ok = (rows <= rr) & (nbytes <= cc * 16)
    print(f'fits {rr} rows x {cc} chunks ({rr * cc} lanes): {ok.mean():.4f} of hot footprints')
h6 = np.abs(Hi[:, 6])
print('|h6| percentiles 50/90/99/max:', np.percentile(h6, [50, 90, 99, 100]))

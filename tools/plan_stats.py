"""Footprint-plan statistics of a workload: how many footprints are single-owner, staged, certified interior."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic, _lib
from meshflow_amd.stabilizer import MeshFlowStabilizer
H, W, F, R, C, omega, iters = 1080, 1920, 300, 16, 16, 10, 100
if len(sys.argv) > 1 and sys.argv[1] == 'cfg3':
    F, R, C, omega, iters = 600, 32, 32, 30, 200
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters)
d_disp = torch.from_numpy(disp).to(dev)
d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
n = 60
table = ops.cell_table(d_disp[:n], d_stab[:n], W, H, R, C)
torch.cuda.synchronize()
buf = table.buf.cpu().numpy()
nrec = n * R * C
plan_off = (nrec * (32 * 8 + 8 + 28 * 4) + 15) & ~15
nfx, nfy = (W + 31) // 32, (H + 7) // 8
npl = n * nfx * nfy
plan = buf[plan_off:plan_off + npl * 16].view(np.uint16).reshape(npl, 8)
reg_off = plan_off + npl * 16                                   # regions follow the plan: 8 bytes per footprint, word 0 = flags | origin
regions = buf[reg_off:reg_off + npl * 8].view(np.uint32)[0::2]
valid = (plan & 0x4000) != 0
overflow = plan[:, 7] == 0xFFFF
ne = np.where(overflow, 9, valid[:, :4].sum(1) + np.where(valid[:, :4].all(1), valid[:, 4:].sum(1), 0))
single = (ne == 1) & ((plan[:, 0] & 0x8000) != 0)
print(f'{npl} footprints: single-owner {single.mean():.3f}; candidates histogram', {int(k): round(float((ne == k).mean()), 4) for k in np.unique(ne)})
codes = plan[:, 4:8]
short = (ne >= 2) & (ne <= 4)
mixed_entries = 0; single_edge = 0
for i in range(4):
    m = short & (i < ne) & ((plan[:, i] & 0x8000) == 0)
    mixed_entries += m.sum(); single_edge += (m & ((codes[:, i] & 7) < 4)).sum()
print(f'MIXED entries in short lists: {mixed_entries}, of which single-edge: {single_edge} ({single_edge / max(mixed_entries, 1):.3f})')
inn = (plan & 0x8000) != 0
cd = codes & 0x3F
pair = (ne == 2) & ~inn[:, 0] & ~inn[:, 1] & (cd[:, 0] < 4) & (cd[:, 1] < 4)
quad = (ne == 4) & ~inn[:, :4].any(1) & ((cd[:, 0] & cd[:, 1] & cd[:, 2] & cd[:, 3] & 8) != 0)
general = ~single & ~pair & ~quad
print(f'paths: single IN {single.mean():.4f}  pair {pair.mean():.4f}  quad {quad.mean():.4f}  general {general.mean():.4f}')
g2 = general & (ne == 2)
print(f'general by shape: ne=1 {(general & (ne == 1)).mean():.4f}  ne=2 {g2.mean():.4f} (of which closed by an IN cell {(g2 & inn[:, 1]).mean():.4f}, '
      f'two-edge codes {(g2 & ~inn[:, 1] & (((cd[:, 0] | cd[:, 1]) & 8) != 0)).mean():.4f})  ne=3 {(general & (ne == 3)).mean():.4f}  '
      f'ne=4 {(general & (ne == 4)).mean():.4f}  ne>4 {(general & (ne > 4)).mean():.4f}')
pairhot = (ne == 2) & ((plan[:, 2] & 0x6000) == 0x2000)          # MF_PLAN_HOT in the unused third entry: the pair path
print(f'plan-certified pair path {pairhot.mean():.4f}')
hot = (plan[:, 1] & 0x6000) == 0x2000                          # MF_PLAN_HOT without VALID: the warp kernel's straight-line path
print(f'staged {((regions >> 31) & 1).mean():.4f}  certified interior {((regions >> 30) & 1).mean():.4f}  hot (single + unit + deep + whole) {hot.mean():.4f}')
regs = buf[reg_off:reg_off + npl * 8].view(np.uint32)[0::2]
print(f'regions: compact {((regs >> 29) & 1).mean():.4f}  staged but not deep {(((regs >> 31) & 1) & ~((regs >> 30) & 1)).mean():.4f}  unstaged {1 - ((regs >> 31) & 1).mean():.4f}')
fast64 = (plan[:, 1] & 0x6002) == 0x2002
print(f'hot with the fast64 certificate {fast64.mean():.4f}')
pf = (ne == 2) & ((plan[:, 2] & 0x6001) == 0x2001)
pv_ = (ne == 2) & ((plan[:, 2] & 0x6003) == 0x2003)
print(f'pair path with the fast certificate {pf.mean():.4f} (of which transposed lanes {pv_.mean():.4f})')
# the warp kernel's own dispatch (warp.hip footprint_body, in its order)
k_hot = (plan[:, 1] & 0x2000) != 0
k_border = ~k_hot & ((regions & 0x08000000) != 0)          # one candidate, several, or none at all (MF_REGION_BORDER)
k_pair = ~k_hot & ~k_border & ((plan[:, 2] & 0x2000) != 0) & ~overflow
k_multi = ~k_hot & ~k_border & ~k_pair & ((plan[:, 4] & 0x2000) != 0) & ~overflow
k_gen = ~(k_hot | k_border | k_pair | k_multi)
print(f'border shapes: one candidate {(k_border & (ne == 1)).mean():.4f}  several {(k_border & (ne >= 2)).mean():.4f}  empty {(k_border & (ne == 0)).mean():.4f}')
print(f'kernel paths: hot {k_hot.mean():.4f}  border {k_border.mean():.4f}  pair {k_pair.mean():.4f}  multi {k_multi.mean():.4f}  general {k_gen.mean():.4f}')
staged = ((regions >> 31) & 1) != 0
deep = ((regions >> 30) & 1) != 0
hist = {int(k): round(float((k_gen & (ne == k)).mean()), 4) for k in np.unique(ne[k_gen])}
print(f'general: by candidates {hist}; staged {(k_gen & staged).mean():.4f}  deep {(k_gen & deep).mean():.4f}  '
      f'one IN cell {(k_gen & single).mean():.4f}  ragged (frame edge) {(k_gen & ~staged).mean():.4f}')
fy = (np.arange(npl) % (nfx * nfy)) // nfx
fxx = np.arange(npl) % nfx
ring = (fy == 0) | (fy == nfy - 1) | (fxx == 0) | (fxx == nfx - 1)
print(f'general on the outermost footprint ring {(k_gen & ring).mean():.4f} (ring = {ring.mean():.4f} of all), second ring '
      f'{(k_gen & ~ring & ((fy == 1) | (fy == nfy - 2) | (fxx == 1) | (fxx == nfx - 2))).mean():.4f}, inside {(k_gen & ~ring & ~((fy == 1) | (fy == nfy - 2) | (fxx == 1) | (fxx == nfx - 2))).mean():.4f}')
gi = k_gen & ~ring & ~((fy == 1) | (fy == nfy - 2) | (fxx == 1) | (fxx == nfx - 2))
print('  inside-general by candidates', {int(k): round(float((gi & (ne == k)).mean()), 4) for k in np.unique(ne[gi])} if gi.any() else {})

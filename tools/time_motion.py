"""Times the vertex-motion kernels (features -> displacements) on a cfg2-sized clip (bench.py reports the CPU port beside them)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic, host, ops

W, H, R, C, F = 1920, 1080, 16, 16, 300
per_pair = tuple(int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (1500, 2500)
_, hom = synthetic.motion(F, R, C, seed=0)
hom[:-1, :2, :2] = np.identity(2) + 0.2 * (hom[:-1, :2, :2] - np.identity(2))
feats = synthetic.features(F, H, W, hom, seed=0, per_pair=per_pair)
early, late, offsets, kmax = host.pack_features(feats)
dev = torch.device('cuda:0')
d = [torch.from_numpy(a).to(dev) for a in (early, late, offsets, np.ascontiguousarray(hom[:-1]))]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    disp, vel, status = ops.vertex_motion(*d, kmax, W, H, R, C, 10, 10)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'GPU: {F - 1} pairs, {early.shape[0]} features (max {kmax}/pair): {dt * 1e3:.2f} ms -> {(F - 1) / dt:.0f} pairs/s')
ops.vertex_motion_check(status)

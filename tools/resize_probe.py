"""Per-launch times of crop + resize in a bench-like context (config-2 frames, the clip's crop rectangle): which buffers, how warm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
H, W, F, R, C = 1080, 1920, 300, 16, 16
dev = torch.device('cuda:0')
d_frames = synthetic.frames_torch(F, H, W, dev, seed=0)
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(device='cuda:0')
d_disp = torch.from_numpy(disp).to(dev)
d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
d_out = torch.empty_like(d_frames)
table = ops.cell_table(d_disp, d_stab, W, H, R, C)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20): ops.warp(d_frames, table, out=d_out)
rect = (13, 11, 1909, 1068)
def series(src, dst, n=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); ops.crop_resize(src, rect, out=dst); b.record()
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b), 3) for a, b in ev]
def block(src, dst, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): ops.crop_resize(src, rect, out=dst)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n, 3)
scratch = torch.empty_like(d_frames)
print('fresh scratch, per launch:', series(d_out, scratch))
print('10 back to back          :', block(d_out, scratch), block(d_out, scratch))
print('into d_frames            :', block(d_out, d_frames))
# what bench.py does before its resize loop: a synchronize, six Jacobi launches (578 wavefronts: most of the chip idle), a synchronize
import time
taps_d, lam_d, inv_on_d = s._jacobi_coefficients_device(F, W, H, 0, hom, dev)
b2d = d_disp.reshape(F, -1); x2d = torch.empty_like(b2d)
for idle_ms in (0, 5, 50):
    torch.cuda.synchronize()
    for _ in range(6): ops.jacobi(b2d, taps_d, lam_d, inv_on_d, 10, 100, out=x2d)
    torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
    print(f'after jacobi x6 + sync + {idle_ms} ms idle, per launch:', series(d_out, scratch, 24))

"""Soak test of the host pipeline's ring (csrc/hostpipe.hip): the same clip through mf_warp_crop_u8c3_host_frames again and again under
random thread / chunk / slot settings, fresh output arrays every time; every result must equal the first byte for byte (and the first one
the device operators').      python tools/soak_hostpipe.py [iterations] [seed]"""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from meshflow_amd import _lib, ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
F, H, W, R, C = 96, 720, 1280, 16, 16
if len(sys.argv) > 3 and sys.argv[3] == 'odd':            # frames of 333 x 251 pixels = 250,749 bytes: every other ring slot starts at an odd address
    F, H, W, R, C = 150, 251, 333, 7, 5
dev = torch.device('cuda:0')
base = synthetic.frames_torch(8, H, W, dev, seed=3).cpu().numpy()
frames = [np.ascontiguousarray(base[i % 8] ^ np.uint8(i)) for i in range(F)]
disp, hom = synthetic.motion(F, R, C, seed=3)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100, device='cuda:0')
stab = np.ascontiguousarray(s._get_stabilized_vertex_displacements(F, frames, 0, disp, hom))
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
border = (ctypes.c_uint8 * 3)(0, 0, 255)
pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in frames])
fb = H * W * 3


def once():
    out = np.empty((F, H, W, 3), np.uint8)
    cr = np.empty((F, H, W, 3), np.uint8)
    crop = np.zeros((F, 4), np.int32)
    bounds = (ctypes.c_int32 * 4)()
    pout = (ctypes.c_void_p * F)(*[out.ctypes.data + i * fb for i in range(F)])
    pcr = (ctypes.c_void_p * F)(*[cr.ctypes.data + i * fb for i in range(F)])
    _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, pout, pcr, p(disp), p(stab), F, W, H, R, C, border, p(crop), bounds, None))
    return out, cr, crop, tuple(bounds)


names = ('MF_PIPE_UP', 'MF_PIPE_DOWN', 'MF_PIPE_POPULATE', 'MF_PIPE_CHUNK', 'MF_PIPE_SLOTS')
out0, cr0, crop0, b0 = once()
# the device operators on the same frames
d_fr = torch.from_numpy(np.stack(frames)).to(dev)
table = ops.cell_table(torch.from_numpy(disp).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
d_out = ops.warp(d_fr, table, (0, 0, 255))
d_crop = table.crop
assert np.array_equal(d_out.cpu().numpy(), out0) and np.array_equal(d_crop.cpu().numpy(), crop0)
sums = (zlib.crc32(out0), zlib.crc32(cr0))
t0 = time.time()
bad = 0
for it in range(iters):
    setting = (int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(0, 9)), int(rng.integers(1, 20)), int(rng.integers(2, 40)))
    for k, v in zip(names, setting):
        os.environ[k] = str(v)
    out, cr, crop, b = once()
    ok = (zlib.crc32(out), zlib.crc32(cr)) == sums and np.array_equal(crop, crop0) and b == b0
    if not ok:
        bad += 1
        print(f'MISMATCH at iteration {it}, setting {setting}', flush=True)
print(f'{iters} runs of {F} x {W}x{H} under random ring settings in {time.time() - t0:.1f} s: {bad} mismatches; rectangle {b0}')
sys.exit(1 if bad else 0)

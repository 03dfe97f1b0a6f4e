"""Phase profile of the warp kernel's hot path (experiment build: make variant NAME=ph EXTRA=-DMF_EXP_PHASES=1).

    python tools/phase_warp.py meshflow_amd/variants/libmf_ph.so [single|cfg2]

Average shader-clock cycles (s_memtime) a hot wavefront spends between its wait points, sampled on every 64th wavefront."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
wl = sys.argv[2] if len(sys.argv) > 2 else 'single'
vp, i = ctypes.c_void_p, ctypes.c_int
lib.mf_cell_table_bytes.restype = ctypes.c_size_t
lib.mf_cell_table_bytes.argtypes = [i] * 5
lib.mf_cell_table_f64.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
lib.mf_warp_u8c3.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp]
lib.mf_debug_phases.argtypes = [vp, i]
dev = torch.device('cuda:0')
H, W, F = 1080, 1920, 300
if wl == 'single':
    R = C = 1
    unstab = np.zeros((F, 2, 2, 2)); stab = unstab.copy(); stab[..., 0] = 3.3; stab[..., 1] = -2.7; stab[:, 1, 1, 0] += 2.0
    d_unstab, d_stab = torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev)
else:
    R = C = 16
    disp, hom = synthetic.motion(F, R, C, seed=0)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100, device='cuda:0')
    d_unstab = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_unstab, W, H, 0, hom)
frames = synthetic.frames_torch(F, H, W, dev, seed=0)
out = torch.empty_like(frames)
table = torch.empty(lib.mf_cell_table_bytes(F, W, H, R, C), dtype=torch.uint8, device=dev)
crop = torch.empty((F, 4), dtype=torch.int32, device=dev)
status = torch.zeros(4, dtype=torch.int32, device=dev)
border = (ctypes.c_uint8 * 3)(0, 0, 255)
st = torch.cuda.current_stream().cuda_stream
assert lib.mf_cell_table_f64(d_unstab.data_ptr(), d_stab.data_ptr(), F, W, H, R, C, table.data_ptr(), crop.data_ptr(), status.data_ptr(), st) == 0
for k in range(3):
    if k == 2:
        torch.cuda.synchronize(); lib.mf_debug_phases(None, 1)
    assert lib.mf_warp_u8c3(frames.data_ptr(), out.data_ptr(), table.data_ptr(), F, W, H, R, C, border, crop.data_ptr(), st) == 0
torch.cuda.synchronize()
h = (ctypes.c_ulonglong * 8)()
lib.mf_debug_phases(h, 0)
n = max(1, h[5])
names = ['entry -> plan + region arrived', 'window copy issued, matrix arrived', 'coordinates (75 f64 ops + 8 fma)', 'wait for the window (vmcnt 0)',
         'taps + blend']
print(f'{wl}: {h[5]} sampled hot wavefronts; s_memtime cycles per phase (each stamp adds its own s_memtime round trip)')
for k, name in enumerate(names):
    print(f'  {name:45s} {h[k] / n:9.0f}')
print(f'  {"sum":45s} {sum(h[:5]) / n:9.0f}')

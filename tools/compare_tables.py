"""Byte-for-byte comparison of the cell table blob (records, boxes, edge sets, PLAN, REGIONS ...) two builds of the library write for the
same clip:   python tools/compare_tables.py cfg2|cfg3|cfg4shard|small libA.so libB.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
wl = sys.argv[1]
H, W, F, R, C, omega, iters = {'cfg2': (1080, 1920, 300, 16, 16, 10, 100), 'cfg3': (1080, 1920, 600, 32, 32, 30, 200),
                               'cfg4shard': (2160, 3840, 150, 16, 16, 10, 100), 'small': (360, 640, 64, 16, 16, 10, 100),
                               'odd': (250, 333, 40, 7, 5, 4, 10)}[wl]
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
d_un = torch.from_numpy(disp).to(dev)
d_st = s._stabilized_vertex_displacements_device(d_un, W, H, 0, hom)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
blobs = []
for path in sys.argv[2:4]:
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.mf_cell_table_bytes.restype = ctypes.c_size_t; lib.mf_cell_table_bytes.argtypes = [i] * 5
    lib.mf_cell_table_f64.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
    tb = lib.mf_cell_table_bytes(F, W, H, R, C)
    buf = torch.zeros(tb, dtype=torch.uint8, device=dev)
    crop = torch.empty((F, 4), dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    assert lib.mf_cell_table_f64(d_un.data_ptr(), d_st.data_ptr(), F, W, H, R, C, buf.data_ptr(), crop.data_ptr(), status.data_ptr(), st) == 0
    torch.cuda.synchronize()
    blobs.append((buf, crop, int(status.item())))
same = torch.equal(blobs[0][0], blobs[1][0]) and torch.equal(blobs[0][1], blobs[1][1]) and blobs[0][2] == blobs[1][2]
print(f'{wl}: {blobs[0][0].numel()} table bytes, identical: {same}')
if not same:
    d = (blobs[0][0] != blobs[1][0]).nonzero().flatten()
    print('  differing bytes:', d.numel(), 'first at', d[:8].tolist())
sys.exit(0 if same else 1)

"""Times mf_cell_table_f64 (cell table + footprint plan) of several library builds, interleaved, warp never run (timing-only variants
may write plans that are not fit for it).  python tools/time_plan.py [cfg2|cfg3] lib1.so lib2.so ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
wl = sys.argv[1]
H, W, F, R, C, omega, iters = {'cfg2': (1080, 1920, 300, 16, 16, 10, 100), 'cfg3': (1080, 1920, 600, 32, 32, 30, 200)}[wl]
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
d_un = torch.from_numpy(disp).to(dev)
d_st = s._stabilized_vertex_displacements_device(d_un, W, H, 0, hom)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = []
for path in sys.argv[2:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.mf_cell_table_bytes.restype = ctypes.c_size_t; lib.mf_cell_table_bytes.argtypes = [i] * 5
    lib.mf_cell_table_f64.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
    tb = lib.mf_cell_table_bytes(F, W, H, R, C)
    libs.append((os.path.basename(path), lib, torch.empty(tb, dtype=torch.uint8, device=dev), torch.empty((F, 4), dtype=torch.int32, device=dev),
                 torch.zeros(1, dtype=torch.int32, device=dev), []))
def run(v):
    assert v[1].mf_cell_table_f64(d_un.data_ptr(), d_st.data_ptr(), F, W, H, R, C, v[2].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), st) == 0
for _ in range(7):
    for v in libs:
        run(v); run(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(v)
        e1.record(); torch.cuda.synchronize()
        v[5].append(e0.elapsed_time(e1) / 10)
for v in libs:
    print(f'{wl} {v[0]:24s} table+plan median {np.median(v[5]) * 1e3:7.1f} us  min {min(v[5]) * 1e3:7.1f}')

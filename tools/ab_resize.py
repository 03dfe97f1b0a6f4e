"""Interleaved A/B of builds of libmeshflow_hip.so on the crop + resize kernel (cfg2 frames, the cfg2 crop rectangle)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
H, W, F = 1080, 1920, 300
dev = torch.device('cuda:0')
frames = synthetic.frames_torch(F, H, W, dev, seed=0)
rect = (13, 11, 1909, 1068)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = []
for p in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(p))
    lib.mf_crop_resize_workspace_bytes.restype = ctypes.c_size_t
    lib.mf_crop_resize_workspace_bytes.argtypes = [ctypes.c_int] * 2
    lib.mf_crop_resize_u8c3.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_void_p]
    libs.append(dict(name=os.path.basename(p), lib=lib, out=torch.empty_like(frames),
                     work=torch.empty(lib.mf_crop_resize_workspace_bytes(W, H), dtype=torch.uint8, device=dev), t=[]))
def run(v):
    rc = v['lib'].mf_crop_resize_u8c3(frames.data_ptr(), v['out'].data_ptr(), F, W, H, *rect, v['work'].data_ptr(), st)
    assert rc == 0
for v in libs:
    run(v)
torch.cuda.synchronize()
for v in libs[1:]:
    print(v['name'], 'identical to', libs[0]['name'], ':', torch.equal(v['out'], libs[0]['out']))
for _ in range(5):
    for v in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(v)
        e1.record(); torch.cuda.synchronize()
        v['t'].append(e0.elapsed_time(e1) / 10)
for v in libs:
    t = np.array(v['t'])
    print(f'{v["name"]:28s} resize median {np.median(t):.4f} ms min {t.min():.4f}  (frac of 8 TB/s {2.0 * H * W * 3 * F / (np.median(t) * 1e-3) / 8e12:.4f})')

"""The single-phase crop pipeline (mf_warp_crop_u8c3_host_frames) under different MF_PIPE_* settings: fresh output array per
call (as stabilize_clip allocates it), separate input frames.   python tools/time_e2e_crop.py sweep|slots|one [cfg2|cfg3|cfg4shard]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

WORKLOADS = {'cfg2': (300, 1080, 1920, 16, 16, 10, 100), 'cfg3': (600, 1080, 1920, 32, 32, 30, 200), 'cfg4shard': (150, 2160, 3840, 16, 16, 10, 100)}
wl = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
if len(sys.argv) > 1 and sys.argv[1] in ('sweep', 'slots', 'grid'):
    # sweep: thread counts x frames per chunk (0 = the default, ~100 MB); slots: ring depth
    grid = [(4, 4, 0, 8), (4, 4, 8, 8), (4, 4, 24, 8), (3, 3, 0, 8), (6, 6, 0, 8), (6, 6, 0, 12), (8, 8, 0, 16), (2, 2, 0, 8)] if sys.argv[1] == 'sweep' else \
           [(4, 4, 0, k) for k in (3, 4, 6, 8, 10, 12, 16)]
    if sys.argv[1] == 'grid':                   # GRID="up,down,chunk,slots;..." in the environment
        grid = [tuple(int(v) for v in item.split(',')) for item in os.environ['GRID'].split(';')]
    for up, down, chunk, slots in grid:
        env = dict(os.environ, MF_PIPE_UP=str(up), MF_PIPE_DOWN=str(down), MF_PIPE_SLOTS=str(slots))
        if chunk:
            env['MF_PIPE_CHUNK'] = str(chunk)
        out = subprocess.run([sys.executable, __file__, 'one', wl], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print(f'{wl} up={up} down={down} chunk={chunk or "default"} slots={slots}:', ' | '.join(out), flush=True)
    sys.exit(0)

from meshflow_amd import _lib, synthetic
F, H, W, R, C, omega, iters = WORKLOADS[wl]
import torch
base = synthetic.frames_torch(4, H, W, torch.device('cuda:0'), seed=0).cpu().numpy()
if os.environ.get('CONTIG'):               # the frames as rows of ONE array (the pipeline merges adjacent frames into one copy per chunk)
    whole = np.empty((F, H, W, 3), np.uint8)
    for i in range(F):
        whole[i] = base[i % 4]
    frames = [whole[i] for i in range(F)]
else:
    frames = [np.ascontiguousarray(base[i % 4]).copy() for i in range(F)]
disp, hom = synthetic.motion(F, R, C, seed=0)
from meshflow_amd.stabilizer import MeshFlowStabilizer
stab = np.ascontiguousarray(MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters,
                                               device='cuda:0')._get_stabilized_vertex_displacements(F, frames, 0, disp, hom))      # the config's own paths
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
border = (ctypes.c_uint8 * 3)(0, 0, 255)
crop = np.zeros((F, 4), np.int32)
fb = H * W * 3
pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in frames])
bounds = (ctypes.c_int32 * 4)()


reused = np.zeros((F, H, W, 3), np.uint8) if os.environ.get('REUSE') else None       # REUSE=1: the output pages exist already


def call(with_crop):
    out = reused if reused is not None else np.empty((F, H, W, 3), np.uint8)
    pout = (ctypes.c_void_p * F)(*[out.ctypes.data + i * fb for i in range(F)])
    ms = ctypes.c_float(0)
    if with_crop:
        _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, None, pout, p(disp), p(stab), F, W, H, R, C, border, p(crop), bounds, ctypes.byref(ms)))
    else:
        _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout, p(disp), p(stab), F, W, H, R, C, border, p(crop), ctypes.byref(ms)))
    return ms.value, out          # (the caller drops `out` OUTSIDE the timed interval: unmapping 1.87 GB costs ~100 ms by itself)


for label, wc in (('warp', False), ('warp+crop', True)):
    t = []
    for _ in range(7):
        t0 = time.perf_counter(); kms, keep = call(wc); t.append(time.perf_counter() - t0); del keep
    t = t[1:]
    print(f'{label}: mean {np.mean(t) * 1e3:.1f} min {np.min(t) * 1e3:.1f} ms ({F / np.mean(t):.0f} fps, {F * fb / np.mean(t) / 1e9:.1f} GB/s each way; kernels {kms:.2f} ms, rectangle {list(bounds)})', flush=True)

"""Where the host-to-host clip time goes (cfg2): the C pipeline on contiguous / per-frame / fresh / reused buffers, thread counts."""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == 'fresh-sweep':
    for pop, thp in ((0, 0), (4, 0), (4, 1), (8, 0), (8, 1), (2, 0)):
        env = dict(os.environ, MF_PIPE_POPULATE=str(pop), MF_PIPE_THP=str(thp))
        out = subprocess.run([sys.executable, __file__, 'fresh'], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print(f'populate={pop} thp={thp}:', ' | '.join(out), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'sweep':
    for up, down, chunk in ((3, 3, 16), (2, 2, 16), (4, 4, 16), (6, 6, 16), (3, 3, 8), (3, 3, 32), (4, 4, 8), (1, 1, 16)):
        env = dict(os.environ, MF_PIPE_UP=str(up), MF_PIPE_DOWN=str(down), MF_PIPE_CHUNK=str(chunk))
        out = subprocess.run([sys.executable, __file__, 'one'], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        print(f'up={up} down={down} chunk={chunk}:', ' | '.join(out), flush=True)
    sys.exit(0)

from meshflow_amd import _lib, synthetic
F, H, W, R, C = 300, 1080, 1920, 16, 16
base = synthetic.frames_numpy(4, H, W, seed=0)
frames = np.ascontiguousarray(np.broadcast_to(base, (F // 4, 4, H, W, 3)).reshape(F, H, W, 3))
disp, hom = synthetic.motion(F, R, C, seed=0)
stab = np.ascontiguousarray(0.3 * disp)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
border = (ctypes.c_uint8 * 3)(0, 0, 255)
crop = np.zeros((F, 4), np.int32)
fb = H * W * 3


def call(ins, out):
    pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in ins])
    pout = (ctypes.c_void_p * F)(*[out.ctypes.data + i * fb for i in range(F)])
    _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout, p(disp), p(stab), F, W, H, R, C, border, p(crop), None))


def bench(label, fn, n=6):
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    t = t[1:]
    print(f'{label}: mean {np.mean(t) * 1e3:.1f} ms min {np.min(t) * 1e3:.1f} ms ({F / np.mean(t):.0f} fps)', flush=True)


out = np.empty_like(frames)
contig = [frames[i] for i in range(F)]
separate = [frames[i].copy() for i in range(F)]
if len(sys.argv) > 1 and sys.argv[1] == 'fresh':
    bench('separate in, fresh out', lambda: call(separate, np.empty_like(frames)))
    sys.exit(0)
bench('contiguous in, reused out', lambda: call(contig, out))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
    sys.exit(0)
bench('separate in, reused out', lambda: call(separate, out))
bench('separate in, fresh out', lambda: call(separate, np.empty_like(frames)))
bench('np.empty + touch 1.87 GB', lambda: np.empty_like(frames).fill(0))
from meshflow_amd.stabilizer import MeshFlowStabilizer
s = MeshFlowStabilizer(device='cuda:0')
import torch
bench("stabilize_clip(list)", lambda: s.stabilize_clip(separate, disp, hom))
bench('stabilize_clip(array)', lambda: s.stabilize_clip(frames, disp, hom))

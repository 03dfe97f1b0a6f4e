"""Does a hipGraph of one clip's kernels (Jacobi sweep -> cell table + plan -> warp) shorten the GPU-side gaps between dependent kernels?
    python tools/graph_probe.py [cfg2|cfg3]
Direct launches on one stream against replays of the captured graph, interleaved rounds, HIP events around N clips each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic, host

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
F, R, C, om, it = {'cfg2': (300, 16, 16, 10, 100), 'cfg3': (600, 32, 32, 30, 200)}[wl]
H, W = 1080, 1920
dev = torch.device('cuda:0')
disp, hom = synthetic.motion(F, R, C, seed=0)
d_disp = torch.from_numpy(disp).to(dev)
taps, lam, inv_on = [torch.from_numpy(a).to(dev) for a in host.jacobi_band_coefficients(F, W, H, 0, hom, om)]
b = d_disp.reshape(F, -1)
stab = torch.empty_like(b)
frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
dst = torch.empty_like(frames)
ops.jacobi(b, taps, lam, inv_on, om, it, out=stab)
table = ops.cell_table(d_disp, stab, W, H, R, C)
torch.cuda.synchronize()


def clip():
    ops.jacobi(b, taps, lam, inv_on, om, it, out=stab)
    ops.cell_table(d_disp, stab, W, H, R, C, table=table)
    ops.warp(frames, table, (0, 0, 255), out=dst)


def sweep():
    ops.jacobi(b, taps, lam, inv_on, om, it, out=stab)


def table_plan():
    ops.cell_table(d_disp, stab, W, H, R, C, table=table)


def warp():
    ops.warp(frames, table, (0, 0, 255), out=dst)


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3): clip()
    torch.cuda.synchronize()
    ref, ref_stab = dst.clone(), stab.clone()
    if len(sys.argv) > 2 and sys.argv[2] == 'trace':          # under rocprofv3 --kernel-trace: ten direct clips, then ten replays of their graph
        for _ in range(10): clip()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            clip()
        torch.cuda.synchronize()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        sys.exit(0)
    def sweep_table(): sweep(); table_plan()
    def table_warp(): table_plan(); warp()
    def warp_sweep(): warp(); sweep()
    for name, fn in (('sweep + table + plan + warp', clip), ('sweep', sweep), ('table + plan', table_plan), ('warp', warp),
                     ('sweep, table + plan', sweep_table), ('table + plan, warp', table_warp), ('warp, sweep', warp_sweep)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
        g.replay(); torch.cuda.synchronize()
        same = bool(torch.equal(dst, ref)) and bool(torch.equal(stab, ref_stab))
        N = 100 if wl == 'cfg2' else 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = {'direct': [], 'graph': []}
        for rnd in range(5):
            for kind in ('direct', 'graph'):
                e0.record()
                for _ in range(N):
                    fn() if kind == 'direct' else g.replay()
                e1.record(); torch.cuda.synchronize()
                res[kind].append(e0.elapsed_time(e1) / N * 1e3)
        print(f'{wl} {name:28s} us per call, one stream, back to back: direct median {np.median(res["direct"]):8.1f} min {min(res["direct"]):8.1f} | graph median {np.median(res["graph"]):8.1f} min {min(res["graph"]):8.1f} | outputs identical: {same}')

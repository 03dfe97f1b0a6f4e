"""The same launch again and again: every result must equal the first byte for byte (rare races -- an in-flight load clobbering a live
register, a missing wait -- show up as one odd run in hundreds).     python tools/determinism_soak.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device('cuda:0')
bad = 0


def soak(tag, runs, fn):
    global bad
    first = fn()
    torch.cuda.synchronize()
    first = [t.clone() for t in first]
    odd = 0
    for _ in range(int(runs * scale)):
        got = fn()
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(got, first)):
            odd += 1
    print(f'{tag}: {int(runs * scale)} runs, {odd} differ from the first', flush=True)
    bad += odd


for name, (F, H, W, R, C, omega, iters), runs in (('640x360 x 64', (64, 360, 640, 16, 16, 10, 100), 1500), ('cfg2', (300, 1080, 1920, 16, 16, 10, 100), 60),
                                                ('cfg3 slice', (120, 1080, 1920, 32, 32, 30, 200), 60), ('333x250 odd', (40, 250, 333, 7, 5, 4, 10), 1500)):
    disp, hom = synthetic.motion(F, R, C, seed=1)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=1)
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    table = ops.CellTable(F, W, H, R, C, dev)
    out = torch.empty_like(d_frames)

    def warp():
        out.fill_(0xEE)
        ops.cell_table(d_disp, d_stab, W, H, R, C, table=table)
        ops.warp(d_frames, table, (0, 0, 255), out=out)
        return out, table.crop, table.buf
    soak(f'{name}: cell table + plan + warp', runs, warp)
    soak(f'{name}: sweep', runs, lambda: (s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom),))
    rect = (W // 40, H // 50, W - 1 - W // 45, H - 1 - H // 60)
    res = torch.empty_like(d_frames)

    def resize():
        res.fill_(0xEE)
        return (ops.crop_resize(out, rect, out=res),)
    soak(f'{name}: crop + resize', runs, resize)

    def scan():
        ops.cell_table(d_disp, d_stab, W, H, R, C, table=table)
        return (ops.crop_scan(table),)
    soak(f'{name}: scan-only', runs, scan)
    soak(f'{name}: stability score', runs, lambda: ops.stability_score(d_stab))
print('odd runs in all:', bad)
sys.exit(1 if bad else 0)

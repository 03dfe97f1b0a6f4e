"""Where should the sweep of clip i + 1 start -- beside clip i's cell table + plan (gate 'table') or behind its plan, beside its warp ('plan')?
`resident_gate = 'auto'` switches at 4 GFLOP of sweep, a threshold set from config 2 (0.8 GFLOP) and config 3 (32.7) alone; this measures
the shapes in between.     python tools/gate_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
dev = torch.device('cuda:0')
H, W = 1080, 1920
for F, R, C, omega, iters in ((300, 16, 16, 10, 100), (300, 32, 32, 10, 100), (300, 16, 16, 30, 200), (600, 16, 16, 30, 100), (300, 32, 32, 30, 100),
                             (600, 16, 16, 30, 200), (300, 32, 32, 30, 200), (600, 32, 32, 30, 200)):
    disp, hom = synthetic.motion(F, R, C, seed=0)
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0)
    d_disp = torch.from_numpy(disp).to(dev)
    out = torch.empty_like(d_frames)
    gflop = iters * F * (R + 1) * (C + 1) * 2 * (4 * omega + 5) / 1e9
    res = {}
    for rep in range(2):
        for gate in ('table', 'plan'):
            s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
            s.resident_gate = gate
            for _ in range(4):
                s.stabilize_resident(d_frames, d_disp, hom, out=out, check='deferred')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(15):
                s.stabilize_resident(d_frames, d_disp, hom, out=out, check='deferred')
            s.finish(); torch.cuda.synchronize()
            res.setdefault(gate, []).append((time.perf_counter() - t0) / 15 * 1e3)
    auto = 'plan' if gflop >= 4 else 'table'
    best = min(res, key=lambda g: min(res[g]))
    print(f'{F} frames, {R}x{C} mesh, omega {omega}, {iters} sweeps = {gflop:5.1f} GFLOP: table {min(res["table"]):.3f} ms  plan {min(res["plan"]):.3f} ms  -> auto picks {auto}, best {best}'
          f'{"" if auto == best or abs(min(res["table"]) - min(res["plan"])) < 0.01 * min(res["plan"]) else "   <-- DISAGREE"}', flush=True)
    del d_frames, out

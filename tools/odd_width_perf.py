"""The warp at widths that are no multiple of 4 (1366 x 768, 854 x 480, ...: no row starts on a dword, so no staged windows -- every footprint takes the
general path) beside their neighbours that are.  End of round 5: 1366 x 768 0.805 -> 0.618 ms per 200 frames with ONE unaligned 12-byte store per lane
instead of twelve byte stores (1368 x 768: 0.50 ms), 1918 x 1080 1.528 -> 1.142 (1920: 0.83).     python tools/odd_width_perf.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshflow_amd import ops, synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer
dev = torch.device('cuda:0')
for W, H in ((1368, 768), (1366, 768), (1365, 768), (1920, 1080), (1918, 1080)):
    F, R, C = 200, 16, 16
    disp, hom = synthetic.motion(F, R, C, seed=0)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100, device='cuda:0')
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0)
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    table = ops.cell_table(d_disp, d_stab, W, H, R, C)
    out = torch.empty_like(d_frames)
    for _ in range(3): ops.warp(d_frames, table, (0, 0, 255), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.warp(d_frames, table, (0, 0, 255), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{W}x{H}: warp {ms:.3f} ms per {F} frames = {2 * F * W * H * 3 / ms / 1e9 / 8000 * 1e3:.3f} of 8 TB/s')

"""Within-process interleaved A/B of several builds of libmeshflow_hip.so on the warp path (cell table + plan, warp kernel).

    python tools/ab_warp.py [--workloads cfg2,cfg3,single] [--rounds 5] [--launches 10] libA.so libB.so ...

The first library is the reference build: every other build's stabilized frames and crop values must be byte-identical
to its output (checked on the device).  Timings are HIP-event times on the launch stream, N variants x M rounds
interleaved in ONE process (median and min per variant), as cdna_hip_programming.md rule 24 asks for deltas below 10 %.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer

WORK = {
    'cfg2': (1080, 1920, 300, 16, 16, 10, 100),
    'cfg3': (1080, 1920, 600, 32, 32, 30, 200),
    'cfg4shard': (2160, 3840, 150, 16, 16, 10, 100),
    'single': (1080, 1920, 300, 1, 1, 0, 0),        # 1x1 mesh: every interior footprint has one IN owner
}


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.mf_cell_table_bytes.restype = ctypes.c_size_t
    lib.mf_cell_table_bytes.argtypes = [i] * 5
    lib.mf_cell_table_f64.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
    lib.mf_warp_u8c3.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp]
    lib.mf_last_error.restype = ctypes.c_char_p
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--workloads', default='cfg2,cfg3')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--launches', type=int, default=10)
    ap.add_argument('--kind', default='pattern')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    libs = [(os.path.basename(p), load(p)) for p in args.libs]
    border = (ctypes.c_uint8 * 3)(0, 0, 255)
    for wl in args.workloads.split(','):
        H, W, F, R, C, omega, iters = WORK[wl]
        if wl == 'single':
            unstab = np.zeros((F, R + 1, C + 1, 2))
            stab = unstab.copy(); stab[..., 0] = 3.3; stab[..., 1] = -2.7; stab[:, 1, 1, 0] += 2.0
            d_unstab, d_stab = torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev)
        else:
            disp, hom = synthetic.motion(F, R, C, seed=0)
            s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega,
                                   optimization_num_iterations=iters, device='cuda:0')
            d_unstab = torch.from_numpy(disp).to(dev)
            d_stab = s._stabilized_vertex_displacements_device(d_unstab, W, H, 0, hom)
        frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind=args.kind)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        state = []
        for name, lib in libs:
            tb = lib.mf_cell_table_bytes(F, W, H, R, C)
            state.append(dict(name=name, lib=lib, table=torch.empty(tb, dtype=torch.uint8, device=dev),
                              crop=torch.empty((F, 4), dtype=torch.int32, device=dev),
                              status=torch.zeros(1, dtype=torch.int32, device=dev), out=torch.empty_like(frames), t_tab=[], t_warp=[]))

        def run_table(v):
            rc = v['lib'].mf_cell_table_f64(d_unstab.data_ptr(), d_stab.data_ptr(), F, W, H, R, C, v['table'].data_ptr(),
                                            v['crop'].data_ptr(), v['status'].data_ptr(), st)
            assert rc == 0, v['lib'].mf_last_error()

        def run_warp(v):
            rc = v['lib'].mf_warp_u8c3(frames.data_ptr(), v['out'].data_ptr(), v['table'].data_ptr(), F, W, H, R, C, border,
                                       v['crop'].data_ptr(), st)
            assert rc == 0, v['lib'].mf_last_error()

        def timed(fn, v, n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(v)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        for v in state:                          # correctness first: same bytes as the first build
            run_table(v); run_warp(v)
            torch.cuda.synchronize()
            assert int(v['status'].item()) == 0
        ref = state[0]
        for v in state[1:]:
            same = torch.equal(v['out'], ref['out']) and torch.equal(v['crop'], ref['crop'])
            print(f'{wl}: {v["name"]} output identical to {ref["name"]}: {same}' +
                  ('' if same else f'  ({int((v["out"] != ref["out"]).sum())} bytes differ)'), flush=True)
        for rnd in range(args.rounds):
            for v in (state if rnd % 2 == 0 else state[::-1]):      # (A B ... then ... B A: position in the round must not matter)
                v['t_tab'].append(timed(run_table, v, max(2, args.launches // 2)))
                run_table(v)
                v['t_warp'].append(timed(run_warp, v, args.launches))
        algo = 2.0 * H * W * 3 * F
        for v in state:
            tw, tt = np.array(v['t_warp']), np.array(v['t_tab'])
            print(f'{wl:9s} {v["name"]:28s} warp median {np.median(tw):.4f} ms  min {tw.min():.4f}  (frac of 8 TB/s {algo / (np.median(tw) * 1e-3) / 8e12:.4f})'
                  f'   table+plan median {np.median(tt):.4f} ms', flush=True)
        del state, frames


if __name__ == '__main__':
    main()

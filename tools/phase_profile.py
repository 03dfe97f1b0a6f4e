"""Where the warp kernel's time goes, at HEAD: per-path-class cost from TIMING-ONLY builds and the life of a hot wavefront from s_memtime
stamps (csrc/warp.hip, MF_EXP_SKIP / MF_EXP_PHASES: experiment builds, nothing of them is in the product library).

    bash tools/phase_profile.sh           builds meshflow_amd/variants/libmf_skip{1,2,4,8,16,31,32,64}.so and libmf_phases.so (hipcc, no GPU needed)
    python tools/phase_profile.py [cfg2|cfg3|cfg4shard ...]      on the GPU box; writes the report to stdout

Per class: the product kernel's time minus the time with the wavefronts of that class returning right after the plan test that selects
them (interleaved in one process, median of the rounds).  skip32: every wavefront returns once its plan and region words have arrived;
skip64: once its window copy is issued as well; skip31: every class returns right after its own test."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from meshflow_amd import synthetic
from meshflow_amd.stabilizer import MeshFlowStabilizer

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = {'cfg2': (1080, 1920, 300, 16, 16, 10, 100), 'cfg3': (1080, 1920, 600, 32, 32, 30, 200), 'cfg4shard': (2160, 3840, 150, 16, 16, 10, 100)}
CLASSES = [(1, 'hot'), (4, 'pair'), (8, 'multi'), (2, 'border'), (16, 'everything else (general code)')]


def load(name):
    path = os.path.join(REPO, 'meshflow_amd', 'libmeshflow_hip.so') if name == 'product' else os.path.join(REPO, 'meshflow_amd', 'variants', f'libmf_{name}.so')
    lib = ctypes.CDLL(path)
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.mf_cell_table_bytes.restype = ctypes.c_size_t
    lib.mf_cell_table_bytes.argtypes = [i] * 5
    lib.mf_cell_table_f64.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp]
    lib.mf_warp_u8c3.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp]
    return lib


def census(table_buf, F, W, H, R, C):
    """Path bits of every footprint, from the plan section of the table blob (layout: csrc/mf_common.h)."""
    nrec = F * R * C
    plan_off = (nrec * (32 * 8 + 8 + (16 + 12) * 4) + 15) & ~15
    nfp = F * ((H + 7) // 8) * ((W + 31) // 32)
    plan = table_buf[plan_off:plan_off + 16 * nfp].view(torch.int32).view(nfp, 4).cpu().numpy().view(np.uint32)
    x, y, z = plan[:, 0], plan[:, 1], plan[:, 2]
    hot = ((x >> 16) & 0x2000) != 0
    border = (((x >> 16) & (0x4000 | 0x1000)) == 0x1000) & ~hot
    pair = ((y & 0x2000) != 0) & ~hot & ~border
    multi = ((z & 0x2000) != 0) & ~hot & ~border & ~pair
    rest = ~(hot | border | pair | multi)
    return {'hot': hot.mean(), 'pair': pair.mean(), 'multi': multi.mean(), 'border': border.mean(), 'everything else (general code)': rest.mean()}, nfp


def main():
    dev = torch.device('cuda:0')
    names = ['product'] + [f'skip{m}' for m in (1, 4, 8, 2, 16, 31, 32, 64)]
    libs = {n: load(n) for n in names}
    phases = load('phases')
    phases.mf_exp_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    border = (ctypes.c_uint8 * 3)(0, 0, 255)
    for wl in (sys.argv[1:] or ['cfg2', 'cfg3']):
        H, W, F, R, C, omega, iters = WORK[wl]
        disp, hom = synthetic.motion(F, R, C, seed=0)
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device='cuda:0')
        d_unstab = torch.from_numpy(disp).to(dev)
        d_stab = s._stabilized_vertex_displacements_device(d_unstab, W, H, 0, hom)
        frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
        out = torch.empty_like(frames)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        tb = libs['product'].mf_cell_table_bytes(F, W, H, R, C)
        table = torch.empty(tb, dtype=torch.uint8, device=dev)
        crop = torch.empty((F, 4), dtype=torch.int32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        assert libs['product'].mf_cell_table_f64(d_unstab.data_ptr(), d_stab.data_ptr(), F, W, H, R, C, table.data_ptr(), crop.data_ptr(), status.data_ptr(), st) == 0
        torch.cuda.synchronize()
        share, nfp = census(table, F, W, H, R, C)

        def warp(lib):
            assert lib.mf_warp_u8c3(frames.data_ptr(), out.data_ptr(), table.data_ptr(), F, W, H, R, C, border, crop.data_ptr(), st) == 0

        def timed(lib, n=10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                warp(lib)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        for lib in libs.values():
            warp(lib); warp(lib)
        t = {n: [] for n in names}
        for rnd in range(6):
            for n in (names if rnd % 2 == 0 else names[::-1]):
                t[n].append(timed(libs[n]))
        med = {n: float(np.median(v)) for n, v in t.items()}
        base = med['product']
        algo = 2.0 * H * W * 3 * F
        print(f'== {wl}: {W}x{H}, {F} frames, {R}x{C} mesh; {nfp} footprints (= wavefronts) per launch; product kernel {base:.4f} ms = {algo / (base * 1e-3) / 8e12:.4f} of 8 TB/s '
              f'(interleaved medians of 6 rounds x 10 launches; boxes of the pool differ by +-4 %)')
        print(f'   {"class":34s} {"footprints":>10s} {"time without it":>16s} {"its cost":>10s} {"of the kernel":>14s} {"ns per wavefront":>17s}')
        for mask, name in CLASSES:
            cost = base - med[f'skip{mask}']
            count = share[name] * nfp
            print(f'   {name:34s} {share[name] * 100:9.2f}% {med[f"skip{mask}"]:13.4f} ms {cost:7.4f} ms {cost / base * 100:12.1f} % {cost * 1e6 / max(count, 1) * 1024 / 8:14.1f}'
                  '   (x 1024 SIMDs / 8 wavefronts per SIMD: the slot time one wavefront of the class adds)')
        print(f'   every class returns after its test (skip31): {med["skip31"]:.4f} ms = {med["skip31"] / base * 100:.1f} % of the kernel: launch + prologue + plan / region round trip + window copy + class tests')
        print(f'   every wavefront returns once plan + region words have arrived (skip32): {med["skip32"]:.4f} ms = {med["skip32"] / base * 100:.1f} %')
        print(f'   ... once the window copy is issued as well (skip64): {med["skip64"]:.4f} ms = {med["skip64"] / base * 100:.1f} %')
        # phases of a hot wavefront
        buf = (ctypes.c_ulonglong * 8)()
        phases.mf_exp_phase_read(buf, 1)
        for _ in range(3):
            warp(phases)
        torch.cuda.synchronize()
        phases.mf_exp_phase_read(buf, 1)
        tp = timed(phases, 5)
        phases.mf_exp_phase_read(buf, 1)
        nwave = max(int(buf[0]), 1)
        p = [buf[k] / nwave for k in range(1, 6)]
        print(f'   life of a HOT wavefront (shader cycles (s_memtime); instrumented kernel {tp:.4f} ms, {nwave} sampled wavefronts):')
        for label, v in zip(('entry -> plan + region words arrived', 'window copy issued, matrix arrived, coordinates + fixed point done', 'wait for the window (s_waitcnt vmcnt(0))', 'taps (48 byte loads) + blend',
                             'store issued -> acknowledged (what s_endpgm waits for)'), p):
            print(f'      {label:70s} {v:10.1f}')
        print(f'      {"sum":70s} {sum(p):10.1f}')
        del frames, out, table


if __name__ == '__main__':
    main()

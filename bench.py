#!/usr/bin/env python3
"""bench.py -- frames/sec of the MeshFlow hot path (Jacobi smoothing + mesh warp) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4shard|small]

A "step" is one pass of the hot path over one synthetic clip whose inputs (frames, vertex
displacements) are already resident in HBM: Jacobi coefficient setup (host, O(F)) -> Jacobi sweep ->
per-cell homography table -> mesh warp + crop scan -> clip-level crop bounds.  Steps are issued back to back
(the host prepares clip i+1 while the GPU works on clip i); the degenerate-mesh counter of all steps is read
once, before the closing barrier.  N = 1 runs
BASELINE.json configs[1] (1080p, 300 frames, 16x16 mesh, 100 Jacobi sweeps, ORIGINAL weights).
N > 1 (launched by torch.distributed.run, one rank per GPU) shards ONE clip of 300*N frames by contiguous
frame range: Jacobi replicated, each rank warps its own 300 frames, one 16-byte all-reduce of the crop
bounds (weak scaling); the RCCL gather of all frames to rank 0 is timed once after the timed region.

Rank 0 prints ONE JSON line (fields: see the task's bench contract) including
  roofline:     warp kernel, algorithmic bytes 2*H*W*3 per frame over its HIP-event time, vs 8 TB/s HBM
  end_to_end:   stabilize_clip() from host frames to host frames (PCIe both ways), N = 1 only; never `value`
  cpu_baseline: the C oracle (oracle/warp_oracle.c, OpenMP) timed on this box's host cores on a bounded
                sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (H, W, frames per GPU, mesh R, mesh C, omega, iters)
    'cfg2': (1080, 1920, 300, 16, 16, 10, 100),
    'cfg3': (1080, 1920, 600, 32, 32, 30, 200),
    'cfg4shard': (2160, 3840, 150, 16, 16, 10, 100),
    'small': (360, 640, 64, 16, 16, 10, 100),
}
HBM_PEAK_BYTES_PER_S = 8.0e12     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def usable_cpus():
    """Host threads this process can really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: t.split()),
                        ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: (t.strip(), open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ('max', '-1'):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(H, W, F, R, C, omega, iters, disp, hom, budget_frames):
    """Time the C oracle on the host: Jacobi for the whole clip + warp of `budget_frames` frames (scaled
    linearly to F frames).  The oracle is only the thing timed here, never part of the product path."""
    from meshflow_amd import synthetic
    from oracle import clib, meshflow_oracle as mo
    # more than 32 threads did not help on the 256-thread hosts measured (tests/cpu_threads.py: 16 -> 60 frames/s,
    # 32 -> 61, 64 -> 53, 256 -> 35), so the baseline is capped there
    threads = min(usable_cpus(), 32, max(budget_frames, 1))
    threads = clib.set_threads(threads)      # the environment may pin OMP_NUM_THREADS (torchrun sets it to 1)
    taps, lam, on = mo.jacobi_band_coefficients(F, W, H, 0, hom, omega)
    b = np.ascontiguousarray(disp.reshape(F, -1))
    t0 = time.perf_counter()
    stab = clib.jacobi_banded(b, taps, lam, np.reciprocal(on), omega, iters, openmp=True).reshape(disp.shape)
    t_jac = time.perf_counter() - t0
    sel = np.linspace(0, F - 1, min(budget_frames, F)).astype(int)
    frames = synthetic.frames_numpy(1, H, W, seed=0, kind='pattern')
    frames = np.ascontiguousarray(np.broadcast_to(frames, (len(sel), H, W, 3)))
    t0 = time.perf_counter()
    clib.warp_clip(frames, R, C, disp[sel], stab[sel], use_bbox=True, openmp=True)
    t_warp = time.perf_counter() - t0
    per_clip = t_jac + t_warp * (F / len(sel))
    return {
        'value': F / per_clip, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
        'sample': f'C oracle (OpenMP, {threads} threads, bbox-culled single-pass warp): Jacobi of the full clip '
                  f'({t_jac:.2f} s) + warp of {len(sel)} of {F} frames ({t_warp:.2f} s), warp scaled linearly',
        'jacobi_s': t_jac, 'warp_s_per_frame': t_warp / len(sel),
    }


def end_to_end(stab, d_frames, disp, hom, F):
    """BASELINE.json's other figure: frames/s of stabilize_clip() from a Python list of NumPy frames in host memory to a list
    of stabilized frames + crop bounds + paths + stability score back in host memory (PCIe both ways, pageable buffers, as
    the reference passes them).  Never `value`.  Best of three."""
    frames = [f.copy() for f in d_frames.cpu().numpy()]           # separate allocations, like a decoder's output
    best = float('inf')
    for _ in range(3):
        t0 = time.perf_counter()
        out = stab.stabilize_clip(frames, disp, hom)
        best = min(best, time.perf_counter() - t0)
        del out
    return {'value': F / best, 'unit': 'frames/s', 'ms_per_clip': best * 1e3,
            'what': 'stabilize_clip(list of F host frames) -> list of F host frames + crop bounds + paths + score; '
                    'chunked, overlapped PCIe staging (meshflow_amd/pipeline.py); best of 3'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--frames-kind', default='pattern', choices=['pattern', 'noise'])
    ap.add_argument('--cpu-frames', type=int, default=300, help='frames warped by the CPU baseline (0 = skip)')
    # default: the whole clip, ~5-10 s of host time
    ap.add_argument('--gather', action='store_true', help='(default for N > 1 in shard mode; kept for compatibility)')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the RCCL gather of all frames to rank 0 after the timed region')
    ap.add_argument('--mode', default='shard', choices=['shard', 'clips'],
                    help='N > 1: "shard" = ONE clip of N x frames sharded by frame range (Jacobi replicated, 16-byte crop '
                         'all-reduce); "clips" = N independent clips, one per GPU, no collective (BASELINE config 5)')
    ap.add_argument('--no-e2e', action='store_true', help='skip the host-buffers-in / host-buffers-out measurement (N = 1 only)')
    ap.add_argument('--as-rank-of', type=int, default=0, metavar='N',
                    help='single process only: do the work rank 0 of an N-GPU "shard" run does (clip of N x frames, own '
                         'frame range, replicated Jacobi) without the collective -- predicts weak scaling on one GPU')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from meshflow_amd import dist as mfdist, host, ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer

    rank, world, device = mfdist.init_from_env('cuda')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    H, W, per_gpu, R, C, omega, iters = WORKLOADS[args.workload]
    clips_mode = args.mode == 'clips' and world > 1
    if clips_mode:                       # every rank owns a whole clip of its own (seed = rank)
        F, lo, hi = per_gpu, 0, per_gpu
    elif args.as_rank_of > 1 and world == 1:
        F = per_gpu * args.as_rank_of
        lo, hi = host.shard_range(F, args.as_rank_of, 0)
    else:
        F = per_gpu * world
        lo, hi = host.shard_range(F, world, rank)

    stab = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega,
                              optimization_num_iterations=iters, device=str(device))
    seed = rank if clips_mode else 0
    disp, hom = synthetic.motion(F, R, C, seed=seed)
    d_disp = torch.from_numpy(disp).to(device)
    d_frames = synthetic.frames_torch(hi - lo, H, W, device, seed=seed, kind=args.frames_kind, first_frame=lo)
    d_out = torch.empty_like(d_frames)
    table = ops.CellTable(hi - lo, W, H, R, C, device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    jev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            jev[i][0].record()
        d_stab = stab._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
        if i is not None:
            jev[i][1].record()
        ops.cell_table(d_disp[lo:hi], d_stab[lo:hi], W, H, R, C, table=table, reset_status=False)
        if i is not None:
            ev[i][0].record()
        ops.warp(d_frames, table, stab.color_outside_image_area_bgr, out=d_out)
        if i is not None:
            ev[i][1].record()
        bounds = ops.crop_reduce(table.crop, W, H)
        if not clips_mode:
            bounds = mfdist.allreduce_crop(bounds)
        return d_stab, bounds              # degenerate-mesh counter accumulates in table.status (checked below)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        d_stab, bounds = step(i)
    table.check()                          # degenerate-mesh check of all K steps: one 4-byte D2H, inside the timing
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        elapsed = float(mfdist.all_reduce_max(t).item())

    warp_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    jac_ms = float(np.mean([a.elapsed_time(b) for a, b in jev]))
    # Jacobi kernel alone (the per-step figure above includes the host-side coefficient set-up), outside the timed region
    taps_d, lam_d, inv_on_d = stab._jacobi_coefficients_device(F, W, H, 0, hom, device)
    b2d = d_disp.reshape(F, -1)
    x2d = torch.empty_like(b2d)
    ops.jacobi(b2d, taps_d, lam_d, inv_on_d, omega, iters, out=x2d)
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    for _ in range(5):
        ops.jacobi(b2d, taps_d, lam_d, inv_on_d, omega, iters, out=x2d)
    k1.record()
    torch.cuda.synchronize()
    jac_kernel_ms = k0.elapsed_time(k1) / 5
    jac_flops = float(iters) * F * b2d.shape[1] * (2 * (2 * omega + 1) + 3)
    # Next row on the path, measured outside the timed region: crop to the clip-level bounds and
    # resize back, device-resident (mfs.py:1111-1157).
    resize_ms = None
    if not (bounds[2] < bounds[0] or bounds[3] < bounds[1]):
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch = torch.empty_like(d_frames)
        rect = bounds.tolist()                       # one D2H of the 4 bounds, outside the timed launches
        ops.crop_resize(d_out, rect, out=scratch)
        r0.record()
        for _ in range(3):
            ops.crop_resize(d_out, rect, out=scratch)
        r1.record()
        torch.cuda.synchronize()
        resize_ms = r0.elapsed_time(r1) / 3
        del scratch
    # The row before the path, also outside the timed region: matched features -> vertex displacements
    # (mfs.py:236-452 after the tracker), synthetic features for the same number of frame pairs.
    motion_row = None
    if rank == 0:
        try:                                          # rank 0 only: a failure here must not strand the other ranks
            from meshflow_amd import host as mfhost
            Fm = min(F, per_gpu)                          # one GPU's worth of frame pairs, whatever N is
            feats = synthetic.features(Fm, H, W, hom[:Fm], seed=seed, per_pair=(1500, 2500))
            early, late, offsets, kmax = mfhost.pack_features(feats)
            d_in = [torch.from_numpy(a).to(device) for a in (early, late, offsets, np.ascontiguousarray(hom[:Fm - 1]))]
            ops.vertex_motion(*d_in, kmax, W, H, R, C, stab.feature_ellipse_row_count, stab.feature_ellipse_col_count)
            m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            m0.record()
            for _ in range(3):
                _, _, status = ops.vertex_motion(*d_in, kmax, W, H, R, C, stab.feature_ellipse_row_count,
                                                 stab.feature_ellipse_col_count)
            m1.record()
            torch.cuda.synchronize()
            ops.vertex_motion_check(status)
            motion_ms = m0.elapsed_time(m1) / 3
            motion_row = {'kernels': 'feature_prep + bitonic sort + vertex_median + median_blur + accumulate',
                          'avg_ms': motion_ms, 'frame_pairs': Fm - 1, 'features': int(early.shape[0]),
                          'pairs_per_s': (Fm - 1) / (motion_ms * 1e-3),
                          'note': 'outside the timed region; latency/ALU bound (no meaningful HBM roofline: 14 MB of features)'}
            del d_in
            if world == 1 and args.cpu_frames > 0:
                from oracle import clib
                threads = clib.set_threads(min(usable_cpus(), 32))
                t1 = time.perf_counter()
                clib.vertex_motion(W, H, R, C, stab.feature_ellipse_row_count, stab.feature_ellipse_col_count, feats, hom[:Fm], openmp=True)
                motion_row['cpu_port_pairs_per_s'] = (Fm - 1) / (time.perf_counter() - t1)
                motion_row['cpu_port_threads'] = threads
        except Exception as e:
            motion_row = {'error': f'{type(e).__name__}: {e}'}
    # north_star's "single RCCL gather over xGMI at the end": timed once, after the timed region (it is 7 x 1.87 GB into
    # one GPU -- an order of magnitude above a step -- and a consumer in host memory is better served by every rank
    # draining its own shard over its own PCIe link).  --no-gather skips it.
    gather_ms, gather_error = None, None
    if world > 1 and not clips_mode and not args.no_gather:
        try:
            barrier()
            t1 = time.perf_counter()
            gathered = mfdist.gather_frames(d_out, F)
            barrier()
            gather_ms = (time.perf_counter() - t1) * 1e3
            del gathered
        except Exception as e:                      # never let the optional collective take the measurement down
            gather_error = f'{type(e).__name__}: {e}'

    if rank == 0:
        algo_bytes = 2.0 * H * W * 3 * (hi - lo)
        achieved = algo_bytes / (warp_ms * 1e-3)
        traffic = None            # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload
        try:
            with open(os.path.join(REPO, 'profiles', 'traffic.json')) as fh:
                traffic = json.load(fh).get(args.workload, {}).get('traffic_bytes')
        except (OSError, ValueError):
            pass
        result = {
            'metric': 'frames/sec stabilize() hot path (Jacobi + mesh warp + crop scan), inputs resident in HBM',
            'value': (F * world if clips_mode else F) * args.steps / elapsed, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'dtype_note': 'float64 vertex paths and pixel coordinates (as the reference), integer fixed-point interpolation on uint8',
            'data': f'synthetic ({args.frames_kind} frames, injected random mesh motion, seed 0)',
            'config': {'workload': f'{args.workload}: {W}x{H}, {per_gpu} frames/GPU ({per_gpu * world} total), {R}x{C} mesh, '
                                   f'omega={omega}, {iters} Jacobi sweeps, ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL',
                       'parallelism': (f'{world} independent clips, one per GPU, no collective' if clips_mode else
                                       f'frame-range shards x{world}, Jacobi replicated, 16-byte crop all-reduce')},
            'roofline': {'kernel': 'warp_kernel', 'bound': 'hbm', 'achieved': achieved / 1e9,
                         'peak': HBM_PEAK_BYTES_PER_S / 1e9, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_BYTES_PER_S,
                         'traffic': traffic, 'traffic_source': 'profiles/traffic.json (PMC: size-resolved TCC_EA0_RDREQ read requests + WRITE_SIZE)'
                         if traffic else None,
                         'algorithmic_bytes_per_launch': algo_bytes, 'avg_launch_ms': warp_ms,
                         'note': 'VALU-issue bound (float64 coordinate arithmetic), not HBM bound: see DESIGN.md'},
            'jacobi': {'avg_ms_incl_host_setup': jac_ms, 'kernel_ms': jac_kernel_ms, 'series': int(d_disp[0].numel()), 'frames': F,
                       'bound': 'fp64 vector ALU + LDS (the state never leaves the chip)', 'achieved': jac_flops / (jac_kernel_ms * 1e-3) / 1e12,
                       'peak': 78.6, 'unit': 'TFLOP/s', 'frac': jac_flops / (jac_kernel_ms * 1e-3) / 78.6e12},
            'crop_bounds': [int(v) for v in bounds.tolist()],
        }
        if resize_ms is not None:
            result['next_rows'] = {'crop_resize': {'kernel': 'resize_kernel', 'avg_launch_ms': resize_ms, 'bound': 'hbm',
                                                   'achieved': algo_bytes / (resize_ms * 1e-3) / 1e9, 'unit': 'GB/s',
                                                   'frac': algo_bytes / (resize_ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
                                                   'note': 'outside the timed region; algorithmic bytes 2*H*W*3 per frame'}}
        if args.as_rank_of > 1 and world == 1:
            result['as_rank_of'] = args.as_rank_of
            result['note'] = (f'PREDICTION, not a measurement of {args.as_rank_of} GPUs: one GPU did rank 0\'s share of a '
                              f'{args.as_rank_of}-GPU run; value = frames of the whole clip / that time')
        if motion_row is not None:
            result.setdefault('next_rows', {})['vertex_motion'] = motion_row
        if gather_ms is not None:
            result['gather_to_rank0_ms'] = gather_ms
            result['gather_note'] = 'one RCCL gather of every rank\'s stabilized frames to rank 0, after the timed region'
        if gather_error is not None:
            result['gather_error'] = gather_error
        if world == 1 and not args.no_e2e and args.as_rank_of <= 1:
            try:
                result['end_to_end'] = end_to_end(stab, d_frames, disp, hom, F)
            except Exception as e:                      # (host memory): never let the side measurement take the line down
                result['end_to_end'] = {'error': f'{type(e).__name__}: {e}'}
        if world == 1 and args.cpu_frames > 0:
            try:
                result['cpu_baseline'] = cpu_baseline(H, W, F, R, C, omega, iters, disp, hom, args.cpu_frames)
            except Exception as e:                      # e.g. no C compiler and no prebuilt oracle on the box
                result['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': 0, 'kind': 'port',
                                          'sample': f'failed: {type(e).__name__}: {e}'}
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

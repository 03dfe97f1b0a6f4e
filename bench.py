#!/usr/bin/env python3
"""bench.py -- frames/sec of the MeshFlow hot path (Jacobi smoothing + mesh warp) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4shard|small] [--mode shard|clips|e2e]

A "step" is one pass of the hot path over one synthetic clip whose inputs (frames, vertex
displacements) are already resident in HBM: Jacobi coefficient setup (host, O(F)) -> Jacobi sweep ->
per-cell homography table -> mesh warp + crop scan -> clip-level crop bounds -> (N > 1) 16-byte crop all-reduce.
It is `meshflow_amd.dist.stabilize_sharded` -- the function tests/test_dist_gloo.py drives under gloo -- with the HIP
operators plugged in.  Steps are issued back to back (the host prepares clip i+1 while the GPU works on clip i); the
degenerate-mesh counter of all steps is read once, before the closing barrier.  N = 1 runs BASELINE.json configs[1]
(1080p, 300 frames, 16x16 mesh, 100 Jacobi sweeps, ORIGINAL weights).

N > 1: one process per GPU.  Either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or by `python bench.py --gpus N` alone: the parent then
starts N fresh child processes with that environment BEFORE anything touches a GPU, waits for them and exits non-zero
if any of them failed.  --mode shard: ONE clip of frames-per-GPU x N frames sharded by contiguous frame range, Jacobi
replicated, one 16-byte all-reduce of the crop bounds (weak scaling); the RCCL gather of all frames to rank 0 that
north_star names is timed once AFTER the timed region, next to the time every rank needs to drain its own shard to host
memory.  --mode clips: N independent clips, no collective (BASELINE config 5).

--mode e2e: `value` / `ms_per_step` are the host-to-host clip -- stabilize_clip() from a Python list of NumPy frames in
host memory to the stabilized frames, crop bounds, paths and score back in host memory (PCIe both ways), mean over K
runs, min beside it.  This is the ">= 500 frames/s end-to-end stabilize()" figure of BASELINE.json; it is never the
default `value`, which is the HBM-resident rate.

Rank 0 prints ONE JSON line (fields: see the task's bench contract) including
  roofline:     warp kernel, algorithmic bytes 2*H*W*3 per frame over its HIP-event time, vs 8 TB/s HBM
  end_to_end:   the e2e figure above for the same workload (mean and min of 5 runs), N = 1 only
  cpu_baseline: the C oracle (oracle/warp_oracle.c, OpenMP) timed on this box's host cores on a bounded sample of the
                same workload (kind "port"), and under "reference_faithful" the reference's own formulation -- dense
                matmul Jacobi per vertex (mfs.py:695-704, 875-876), per-cell full-frame painter loop (mfs.py:1031-1061)
                -- restated in NumPy, timed on a few vertices / frames / cells and scaled (N = 1 only)
  cfg1:         BASELINE.json configs[0] (videos/video-1/video-1.m4v through the reference on the CPU) needs a video
                decoder; reported as "skipped: no decoder" where cv2 is absent.
"""
import argparse
import json
import os
import socket
import subprocess
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
HBM_ACHIEVABLE_BYTES_PER_S = 6.29e12   # ... the float4-copy rate of that guide: what a pure streaming kernel reaches
PCIE_LINK_GBPS = 63.0             # MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s per direction (spec) -- the CEILING of the host path


def hbm_roofline_extras(algo_bytes, event_ms, workload):
    """What goes beside `frac` in a warp roofline: the fraction of the ACHIEVABLE copy rate, and the kernel's average duration in the last
    rocprofv3 kernel trace kept under profiles/ (profiles/kernel_trace.json, tools/make_trace_json.py) next to this run's HIP-event time."""
    extra = {'peak_achievable': HBM_ACHIEVABLE_BYTES_PER_S / 1e9, 'frac_of_achievable': algo_bytes / (event_ms * 1e-3) / HBM_ACHIEVABLE_BYTES_PER_S,
             'event_avg_launch_ms': event_ms}
    try:
        with open(os.path.join(REPO, 'profiles', 'kernel_trace.json')) as fh:
            tr = json.load(fh).get(workload)
        if tr and tr.get('algorithmic_bytes_per_launch') == algo_bytes:
            extra['trace'] = {'avg_launch_ms': tr['avg_ms'], 'frac': algo_bytes / (tr['avg_ms'] * 1e-3) / HBM_PEAK_BYTES_PER_S, 'launches': tr['calls'],
                              'source': tr['source'], 'note': 'rocprofv3 --kernel-trace --stats of bench.py on another box of the pool, committed: '
                                                              'profiled runs clock lower and the boxes differ by +-4 %'}
    except (OSError, ValueError, KeyError):
        pass
    return extra


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
    }, stab


def cpu_reference_faithful(H, W, F, R, C, omega, iters, disp, hom, stab, vertices=8, frames=2, cells=16):
    """The reference's OWN formulation of the two stages, restated in NumPy (oracle/meshflow_oracle.py) and timed on a
    bounded sample, scaled linearly (CPU baseline figure 1):
      Jacobi: dense F x F coefficient matrices (mfs.py:713-783) and, per vertex, `iters` x two dense np.matmul
              (mfs.py:695-704, 871-876) on `vertices` of the V vertices;
      warp:   per frame the row-major painter loop over cells -- two 4-point homographies, the full-frame float64 mask
              through the bilinear warpPerspective, perspectiveTransform of every pixel, two np.where (mfs.py:1031-1061)
              -- on `cells` of the R*C cells of `frames` frames, plus the per-frame remap and four edge scans
              (mfs.py:1063-1098).  Every cell does the same full-frame passes, so the loop scales with the cell count.
    OpenCV's own C++ would be faster than these NumPy restatements of its kernels; the figure is an upper bound on the
    reference's time, labelled as scaled."""
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    V = (R + 1) * (C + 1)
    vertices = min(vertices, V)
    cells = min(cells, R * C)
    frames = min(frames, F)
    t0 = time.perf_counter()
    off, on = mo.jacobi_method_input(F, W, H, 0, hom, omega)
    t_setup = time.perf_counter() - t0
    flat = disp.reshape(F, V, 2)
    vsel = np.unique(np.linspace(0, V - 1, vertices).astype(int))
    t0 = time.perf_counter()
    for v in vsel:
        mo.jacobi_method_output_dense(off, on, flat[:, v], flat[:, v], iters)
    t_vertex = (time.perf_counter() - t0) / len(vsel)
    t_jac = t_setup + t_vertex * V
    fsel = np.linspace(0, F - 1, frames).astype(int)
    frame = synthetic.frames_numpy(1, H, W, seed=0, kind='pattern')[0]
    t_fixed = t_cells = 0.0
    for f in fsel:
        t0 = time.perf_counter()
        mo.warp_frame(frame, R, C, disp[f], stab[f], max_cells=0)
        t1 = time.perf_counter()
        mo.warp_frame(frame, R, C, disp[f], stab[f], max_cells=cells, full_mask=True)
        t2 = time.perf_counter()
        t_fixed += t1 - t0
        t_cells += max((t2 - t1) - (t1 - t0), 0.0)
    per_frame = t_fixed / len(fsel) + (t_cells / len(fsel)) * (R * C / cells)
    per_clip = t_jac + per_frame * F
    return {
        'value': F / per_clip, 'unit': 'frames/s', 'cores': 1,
        'kind': f'reference-faithful, scaled from {len(vsel)} of {V} vertices and {cells} of {R * C} cells of {len(fsel)} of {F} frames',
        'sample': f'NumPy restatement of the reference\'s formulation (single thread apart from BLAS): dense Jacobi set-up {t_setup:.2f} s + '
                  f'{t_vertex * 1e3:.1f} ms per vertex; painter loop {t_cells / len(fsel) / cells * 1e3:.0f} ms per cell and frame + '
                  f'{t_fixed / len(fsel):.2f} s per frame for templates, remap and edge scans',
        'jacobi_s_per_clip': t_jac, 'warp_s_per_frame': per_frame,
    }


def cfg1_status():
    """BASELINE.json configs[0]: videos/video-1/video-1.m4v through the reference's CPU path (mfs.py:1325-1340).  Needs a
    video decoder (cv2); the reference and its videos do not exist on the GPU box either."""
    try:
        import cv2  # noqa: F401
    except Exception:
        return 'skipped: no decoder'
    video = os.path.join(os.environ.get('MESHFLOW_REFERENCE_DIR', '/root/reference'), 'videos', 'video-1', 'video-1.m4v')
    if not os.path.exists(video):
        return 'skipped: videos/video-1/video-1.m4v is not on this box'
    return 'skipped: decoder present but the CPU reference leg is not part of this bench (run the reference directly)'


def host_clip(stab, frames, disp, hom, runs, **kw):
    """stabilize_clip() from host frames to host frames, `runs` times: list of seconds."""
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        out = stab.stabilize_clip(frames, disp, hom, **kw)
        times.append(time.perf_counter() - t0)
        del out
    return times


def end_to_end(stab, d_frames, disp, hom, F, runs=5):
    """BASELINE.json's other figure: frames/s of stabilize_clip() from a Python list of NumPy frames in host memory to a list
    of stabilized frames + crop bounds + paths + stability score back in host memory (PCIe both ways, pageable buffers, as
    the reference passes them).  Never `value` outside --mode e2e.  One untimed run first, then mean and min of `runs`.
    Two variants: the path up to mfs.py:158 (stabilized frames back), and -- `with_crop` -- what stabilize() hands its encoder:
    mfs.py:150-162 including _crop_frames (mfs.py:159), only the cropped + resized frames travelling back."""
    frames = [f.copy() for f in d_frames.cpu().numpy()]           # separate allocations, like a decoder's output
    res = {}
    clip_bytes = float(d_frames.numel())
    try:
        pcie = pcie_probe(d_frames.device)
    except Exception as e:                                       # (host memory for the pinned gigabytes)
        pcie = {'error': f'{type(e).__name__}: {e}'}

    def pcie_roofline(seconds):
        return pcie_link_roofline(clip_bytes, seconds, pcie)
    for key, kw in (('', {}), ('with_crop', {'crop': True, 'keep_uncropped': False})):
        host_clip(stab, frames, disp, hom, 1, **kw)
        t = host_clip(stab, frames, disp, hom, runs, **kw)
        mean, best = float(np.mean(t)), float(np.min(t))
        r = {'value': F / mean, 'unit': 'frames/s', 'ms_per_clip': mean * 1e3, 'min_ms_per_clip': best * 1e3, 'best_value': F / best, 'runs': runs,
             'steps': runs, 'warmup': 1, 'ms_per_step': mean * 1e3, 'elapsed_s': float(np.sum(t)),
             'roofline': pcie_roofline(mean)}
        if key:
            r['what'] = ('stabilize_clip(crop=True, keep_uncropped=False): the same + clip-level crop rectangle + _crop_frames (mfs.py:159) on the '
                         'device in the same pipeline (mf_warp_crop_u8c3_host_frames); only the cropped + resized frames come back')
            res[key] = r
        else:
            r['what'] = ('stabilize_clip(list of F host frames) -> list of F stabilized host frames + crop bounds + paths + score; chunked, '
                         f'overlapped PCIe staging below Python (csrc/hostpipe.hip, mf_warp_u8c3_host_frames); mean of {runs} runs after one '
                         'untimed run, min beside it')
            res.update(r)
    res['pcie_probe'] = pcie
    # The ceiling of the pageable path: the same clip through the C ABI with buffers from mf_malloc_host (pinned: the copies are
    # truly asynchronous, no staging by the runtime), raw ctypes.
    try:
        import ctypes
        from meshflow_amd import _lib
        lib = _lib.lib
        n, H, W = len(frames), frames[0].shape[0], frames[0].shape[1]
        nbytes = n * H * W * 3
        hin, hout = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.mf_malloc_host(ctypes.byref(hin), nbytes))
        _lib.check(lib.mf_malloc_host(ctypes.byref(hout), nbytes))
        try:
            fb = H * W * 3
            for i, f in enumerate(frames):
                ctypes.memmove(hin.value + i * fb, f.ctypes.data, fb)
            unstab = np.ascontiguousarray(disp, dtype=np.float64)
            stab_paths = stab._get_stabilized_vertex_displacements(n, frames, 0, disp, hom)
            crop = np.zeros((n, 4), np.int32)
            border = (ctypes.c_uint8 * 3)(*[int(c) for c in stab.color_outside_image_area_bgr])
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            t = []
            with __import__('torch').cuda.device(d_frames.device):
                for k in range(4):
                    t0 = time.perf_counter()
                    _lib.check(lib.mf_warp_u8c3_host(hin, hout, p(unstab), p(np.ascontiguousarray(stab_paths)), n, W, H, stab.mesh_row_count,
                                                     stab.mesh_col_count, border, p(crop), None))
                    t.append(time.perf_counter() - t0)
            t = t[1:]
            res['pinned_buffers'] = {'value': F / float(np.mean(t)), 'unit': 'frames/s', 'ms_per_clip': float(np.mean(t)) * 1e3, 'runs': len(t),
                                     'roofline': pcie_roofline(float(np.mean(t))),
                                     'what': 'mf_warp_u8c3_host on buffers from mf_malloc_host (pinned; warp stage only: paths given).  NOT a ceiling of the pageable '
                                             'path on these boxes: pinned copies all go through the SDMA engines, the runtime\'s staged pageable copies do not'}
        finally:
            lib.mf_free_host(hin)
            lib.mf_free_host(hout)
    except Exception as e:
        res['pinned_buffers'] = {'error': f'{type(e).__name__}: {e}'}
    return res


def pcie_link_roofline(bytes_each_way, seconds, probe):
    """Every frame goes up once and comes down once, both directions at the same time: the ceiling is the LINK's rate per direction
    (PCIe Gen5 x16: 63 GB/s, MI355X_MICROARCH.md).  What hipMemcpyAsync reaches on this box -- one direction alone, both at once -- is
    context, not the peak: a probe taken minutes apart on a shared host has come out BELOW the clip's own rate."""
    ach = bytes_each_way / seconds / 1e9
    probe = probe or {}
    one = [v for v in (probe.get('h2d_alone_GBps'), probe.get('d2h_alone_GBps')) if v]
    return {'bound': 'pcie', 'achieved': ach, 'peak': PCIE_LINK_GBPS, 'unit': 'GB/s per direction, both directions busy', 'frac': ach / PCIE_LINK_GBPS,
            'bytes_each_way': bytes_each_way, 'probe_one_direction_alone': max(one) if one else None,
            'probe_both_directions': probe.get('both_GBps_per_direction'),
            'note': 'peak = the link (PCIe Gen5 x16, 63 GB/s per direction); the probes are this box\'s hipMemcpyAsync rates on 1 GiB of pinned memory, for context'}


def pcie_probe(device, nbytes=1 << 30, reps=3):
    """What this box's PCIe link gives pinned host memory: hipMemcpyAsync of `nbytes` host->device and device->host, each alone and both
    at once (GB/s per direction, best of `reps`), as ONE copy per direction and cut into 64 MiB pieces over four streams per direction
    (how csrc/hostpipe.hip moves a clip).  The roofline of the host-to-host figures -- every frame goes up once and comes down once,
    concurrently -- is the best both-directions rate of the two."""
    import torch
    pin_up = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    pin_dn = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    d_up = torch.empty(nbytes, dtype=torch.uint8, device=device)
    d_dn = torch.empty(nbytes, dtype=torch.uint8, device=device)
    pin_up.fill_(1)
    ups = [torch.cuda.Stream(device=device) for _ in range(4)]
    dns = [torch.cuda.Stream(device=device) for _ in range(4)]
    piece = 64 << 20

    def run(up, down, streams):
        best = 1e9
        for _ in range(reps + 1):                                # (the first pass also warms the path up: best-of)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if streams == 1:
                if up:
                    with torch.cuda.stream(ups[0]):
                        d_up.copy_(pin_up, non_blocking=True)
                if down:
                    with torch.cuda.stream(dns[0]):
                        pin_dn.copy_(d_dn, non_blocking=True)
            else:
                for i, off in enumerate(range(0, nbytes, piece)):
                    if up:
                        with torch.cuda.stream(ups[i % streams]):
                            d_up[off:off + piece].copy_(pin_up[off:off + piece], non_blocking=True)
                    if down:
                        with torch.cuda.stream(dns[i % streams]):
                            pin_dn[off:off + piece].copy_(d_dn[off:off + piece], non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return nbytes / best / 1e9
    res = {'h2d_alone_GBps': run(True, False, 1), 'd2h_alone_GBps': run(False, True, 1),
           'both_one_copy_each_GBps_per_direction': run(True, True, 1), 'both_4_streams_64MiB_pieces_GBps_per_direction': run(True, True, 4),
           'bytes': nbytes, 'note': 'pinned host memory, hipMemcpyAsync; best of 3'}
    res['both_GBps_per_direction'] = max(res['both_one_copy_each_GBps_per_direction'], res['both_4_streams_64MiB_pieces_GBps_per_direction'])
    del pin_up, pin_dn, d_up, d_dn
    return res


def frame_checksums(frames):
    """One position-weighted 63-bit sum per frame of a (n, H, W, 3) uint8 device tensor (tests compare runs with it)."""
    import torch
    n = frames.shape[0]
    words = frames.reshape(n, -1)
    weights = (torch.arange(words.shape[1], device=frames.device, dtype=torch.int64) % 65521) + 1
    return [int(((words[i].to(torch.int64) * weights).sum() & 0x7FFFFFFFFFFFFFFF).item()) for i in range(n)]


def visible_gpus():
    """GPUs of this node WITHOUT touching torch.cuda or HIP (the launcher parent must not initialise the GPU before it starts its
    children): the KFD topology lists one node per agent, GPUs are the ones with SIMDs.  None when /sys cannot tell."""
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        count = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get('simd_count', '0')) > 0:
                count += 1
        return count
    except (OSError, ValueError):
        return None if os.path.exists('/dev/kfd') else 0          # no kernel driver at all: no GPU


def launch_children(args, script=None, argv=None, check_gpus=True):
    """`python bench.py --gpus N` by itself: N fresh child processes, one per GPU, with torchrun's environment.  Nothing in
    this parent touches a GPU, torch.cuda or HIP (the device count comes from /sys, else --gpus is trusted); children are separate
    processes started with subprocess -- never an exec of a process that has initialised the GPU.  Exit code: the first non-zero
    child code; when a rank dies the others are terminated by exact PID, and killed if they ignore that for 10 s (a rank stuck in a
    collective may).  (script / argv / check_gpus: tests/test_bench_launcher.py starts eight stand-in ranks on a box without GPUs.)"""
    backend = os.environ.get('MESHFLOW_DIST_BACKEND', 'nccl')
    visible = visible_gpus() if check_gpus else None
    if visible is not None and (visible < 1 or (backend == 'nccl' and visible < args.gpus)):
        print(f'bench.py: --gpus {args.gpus} but {visible} GPU(s) visible (one rank per GPU under RCCL; '
              f'MESHFLOW_DIST_BACKEND=gloo lets ranks share a GPU for functional tests)', file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        # (the host driver of this pool only supports dmabuf IPC: without this RCCL and cross-process tensor sharing fail with
        # "hipIpcGetMemHandle: invalid argument" -- the build environment's own note; an operator's setting wins)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cpus() // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv)), env=env))
    rc = 0
    pending = list(procs)
    deadline = None                              # set when the survivors have been asked to stop
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:              # a dead rank strands the others in their collectives: end them (exact PIDs)
                    q.terminate()
                deadline = time.monotonic() + 10.0
        if deadline is not None and pending and time.monotonic() > deadline:
            for q in pending:
                q.kill()
            deadline = time.monotonic() + 1e9
        time.sleep(0.05)
    return rc


class KernelPath:
    """The timed step: K passes of the PRODUCT's device-resident pipeline through its PUBLIC method, `MeshFlowStabilizer.stabilize_resident`
    (no private stages; the degenerate-mesh check opted into its DEFERRED form, `finish()` once inside the timed region) -- host coefficient set-up, Jacobi
    sweep on the stabilizer's prep stream, cell table + plan, the warp alone, the clip rectangle folded together by the kernel (-> 16-byte
    all-reduce at N > 1), inputs resident in HBM, clips issued back to back.  HIP events bracket the warp kernel on the caller's stream and
    the sweep stage on the prep stream (the streams they are launched on).  `serial`: the same kernels in order on ONE stream (private
    stages handed to dist.stabilize_sharded: the figure comparable with rounds 1-2)."""

    def __init__(self, stab, d_frames, d_disp, hom, F, frame_range, W, H, R, C, device, collective=False, no_events=False):
        import torch
        self.stab, self.d_frames, self.d_disp, self.hom, self.F, self.range = stab, d_frames, d_disp, hom, F, frame_range
        self.W, self.H, self.R, self.C, self.device, self.collective, self.no_events = W, H, R, C, device, collective, no_events
        self.d_out = torch.empty_like(d_frames)
        self.inputs_ready = torch.cuda.Event()
        self.inputs_ready.record(torch.cuda.current_stream(device))
        self.serial_table = {}

    def step(self, events=None):
        """One clip through the public method."""
        we, je = events if events else (None, None)
        _, bounds, d_stab = self.stab.stabilize_resident(self.d_frames, self.d_disp, self.hom, out=self.d_out, frame_range=self.range,
                                                         inputs_ready=self.inputs_ready, collective=self.collective,
                                                         warp_events=we, jacobi_events=je, check='deferred')
        return d_stab, bounds

    def serial_step(self, events=None):
        from meshflow_amd import dist as mfdist, ops
        stab, W, H, R, C = self.stab, self.W, self.H, self.R, self.C
        we, je = events if events else (None, None)

        def jacobi_fn():
            if je:
                je[0].record()
            d_stab = stab._stabilized_vertex_displacements_device(self.d_disp, W, H, 0, self.hom)
            if je:
                je[1].record()
            return d_stab

        def warp_fn(lo_, hi_, d_stab):
            if 't' not in self.serial_table:
                self.serial_table['t'] = ops.CellTable(hi_ - lo_, W, H, R, C, self.device)
            table = self.serial_table['t']
            ops.cell_table(self.d_disp[lo_:hi_], d_stab[lo_:hi_], W, H, R, C, table=table, reset_status=False)
            if we:
                we[0].record()
            ops.warp(self.d_frames, table, stab.color_outside_image_area_bgr, out=self.d_out)
            if we:
                we[1].record()
            return self.d_out, table

        _, bounds, d_stab, _ = mfdist.stabilize_sharded(self.F, jacobi_fn, warp_fn, lambda table: ops.crop_reduce(table.crop, W, H),
                                                        frame_range=self.range, collective=self.collective)
        return d_stab, bounds

    def finish(self, serial):
        if serial:
            if 't' in self.serial_table:
                self.serial_table['t'].check()
        else:
            self.stab.finish()                # the deferred degenerate-mesh verdicts of the last clips: inside the timed region

    def measure(self, steps, warmup, serial, barrier, max_over_ranks):
        import torch
        step = self.serial_step if serial else self.step
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        jev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        # Clock spin-up, then the W warm-up steps, then the K timed ones back to back.  After an idle period (set-up: milliseconds) this
        # GPU runs the launches of the following ~2-15 ms 5-25 % slower than in steady state (tools/resize_probe.py shows it launch by
        # launch; MF_BENCH_PER_STEP=1 shows it here) -- a power-management transient, not a property of the kernels.  ~25 ms of the same
        # step in front of the warm-up take it out of the timed region; the timed region is still exactly K steps.
        n = self.range[1] - self.range[0]
        spinup_steps = min(50, int(np.ceil(25e-3 / max(2.0 * n * self.H * self.W * 3 / 3.2e12, 1e-4))))
        for _ in range(spinup_steps + warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            d_stab, bounds = step(None if self.no_events else (ev[i], jev[i]))
        self.finish(serial)                    # degenerate-mesh check of all K steps, inside the timing
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        return {'elapsed': elapsed, 'spinup_steps': spinup_steps, 'bounds': bounds.clone(), 'd_stab': d_stab, 'ev': ev, 'jev': jev}

    def latency(self):
        import torch
        lat = []
        for _ in range(7):
            self.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            self.step()
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        self.stab.finish()
        return {'median': float(np.median(lat)), 'min': float(np.min(lat)),
                'note': 'one clip through the product pipeline between two synchronisations, right after another clip '
                        '(clocks up): host issue + Jacobi sweep + first table + warp chunks; median / min of 7'}

    def serial(self, steps, warmup):
        import torch
        for _ in range(warmup):
            self.serial_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            self.serial_step()
        self.finish(True)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / steps * 1e3
        return {'ms_per_step': serial_ms, 'value': self.F / (serial_ms * 1e-3), 'unit': 'frames/s',
                'note': 'the same K steps with every kernel of a clip in order on ONE stream (Jacobi -> cell table + plan -> warp + scan '
                        '-> reduce), no overlap inside or across clips: the figure comparable with rounds 1-2'}

    def warp_alone_ms(self, d_stab, reps=8):
        """The warp kernel with NOTHING beside it (cell table + plan once, then `reps` launches between two events), outside any timed
        region: what the in-pipeline launch time is to be read against when a long sweep shares the chip with the warp (gate 'plan')."""
        import torch
        from meshflow_amd import ops
        lo, hi = self.range
        table = ops.cell_table(self.d_disp[lo:hi], d_stab[lo:hi], self.W, self.H, self.R, self.C)
        for _ in range(3):
            ops.warp(self.d_frames, table, self.stab.color_outside_image_area_bgr, out=self.d_out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ops.warp(self.d_frames, table, self.stab.color_outside_image_area_bgr, out=self.d_out)
        b.record()
        torch.cuda.synchronize()
        table.check()
        return a.elapsed_time(b) / reps

    def jacobi_kernel_ms(self, omega, iters, reps=5):
        """The sweep kernel alone (the per-step figure includes the host-side coefficient set-up), outside any timed region."""
        import torch
        from meshflow_amd import ops
        taps_d, lam_d, inv_on_d = self.stab._jacobi_coefficients_device(self.F, self.W, self.H, 0, self.hom, self.device)
        b2d = self.d_disp.reshape(self.F, -1)
        x2d = torch.empty_like(b2d)
        ops.jacobi(b2d, taps_d, lam_d, inv_on_d, omega, iters, out=x2d)
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for _ in range(reps):
            ops.jacobi(b2d, taps_d, lam_d, inv_on_d, omega, iters, out=x2d)
        k1.record()
        torch.cuda.synchronize()
        ms = k0.elapsed_time(k1) / reps
        flops = float(iters) * self.F * b2d.shape[1] * (2 * (2 * omega + 1) + 3)
        return ms, flops, int(b2d.shape[1])


def other_workload(name, device, steps, pcie):
    """A second configuration in the SAME run (N = 1): its kernel path through the public method -- warp launch time and roofline
    fraction, step time, Jacobi kernel -- and its host-to-host clip with _crop_frames against the PCIe rate measured in this run.
    Bounded to a few seconds; `--no-workloads` skips it."""
    import torch
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    H, W, F, R, C, omega, iters = WORKLOADS[name]
    stab = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters, device=str(device))
    disp, hom = synthetic.motion(F, R, C, seed=0)
    d_disp = torch.from_numpy(disp).to(device)
    d_frames = synthetic.frames_torch(F, H, W, device, seed=0, kind='pattern')
    kp = KernelPath(stab, d_frames, d_disp, hom, F, (0, F), W, H, R, C, device)
    sync = torch.cuda.synchronize
    m = kp.measure(steps, 3, False, sync, lambda t: t)
    warp_ms = float(np.mean([a.elapsed_time(b) for a, b in m['ev']]))
    jac_ms, jac_flops, series = kp.jacobi_kernel_ms(omega, iters)
    gate = stab._sweep_gate(d_disp)
    alone_ms = kp.warp_alone_ms(m['d_stab'])
    algo = 2.0 * H * W * 3 * F
    res = {'config': f'{W}x{H}, {F} frames, {R}x{C} mesh, omega={omega}, {iters} Jacobi sweeps', 'steps': steps,
           'ms_per_step': m['elapsed'] / steps * 1e3, 'value': F * steps / m['elapsed'], 'unit': 'frames/s',
           'warp': {'avg_launch_ms': warp_ms, 'algorithmic_bytes_per_launch': algo, 'achieved': algo / (warp_ms * 1e-3) / 1e9, 'unit': 'GB/s',
                    'frac': algo / (warp_ms * 1e-3) / HBM_PEAK_BYTES_PER_S, **hbm_roofline_extras(algo, warp_ms, name),
                    'beside_the_warp': ('the next clip\'s Jacobi sweep (prep stream, gate "plan": a sweep this long is cheaper beside the warp than beside '
                                        'cell table + plan)' if gate == 'plan' else 'nothing'),
                    'alone': {'avg_launch_ms': alone_ms, 'frac': algo / (alone_ms * 1e-3) / HBM_PEAK_BYTES_PER_S,
                              'frac_of_achievable': algo / (alone_ms * 1e-3) / HBM_ACHIEVABLE_BYTES_PER_S,
                              'note': 'the same launch with nothing beside it, 8 launches between two events outside the timed steps'}},
           'jacobi': {'kernel_ms': jac_ms, 'achieved': jac_flops / (jac_ms * 1e-3) / 1e12, 'unit': 'TFLOP/s', 'frac': jac_flops / (jac_ms * 1e-3) / 78.6e12,
                      'series': series},
           'crop_bounds': [int(v) for v in m['bounds'].tolist()]}
    frames = [f.copy() for f in d_frames.cpu().numpy()]
    del kp, d_frames
    kw = {'crop': True, 'keep_uncropped': False}
    host_clip(stab, frames, disp, hom, 1, **kw)
    t = host_clip(stab, frames, disp, hom, 3, **kw)
    mean = float(np.mean(t))
    res['end_to_end'] = {'value': F / mean, 'unit': 'frames/s', 'ms_per_clip': mean * 1e3, 'min_ms_per_clip': float(np.min(t)) * 1e3, 'runs': 3,
                         'what': 'stabilize_clip(crop=True, keep_uncropped=False), pageable NumPy frame list in, list out',
                         'roofline': pcie_link_roofline(float(F) * H * W * 3, mean, pcie)}
    return res


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--frames-kind', default='pattern', choices=['pattern', 'noise'])
    ap.add_argument('--frames', type=int, default=0, help='frames per GPU instead of the workload\'s own count (a SLICE of the clip: '
                    'profiling aid, e.g. the cfg3 PMC passes; the line says so)')
    ap.add_argument('--cpu-frames', type=int, default=300, help='frames warped by the CPU baseline (0 = skip both CPU legs)')
    # default: the whole clip, ~5-10 s of host time
    ap.add_argument('--no-faithful', action='store_true', help='skip the reference-faithful (dense / per-cell) CPU leg (~20 s)')
    ap.add_argument('--gather', action='store_true', help='(default for N > 1 in shard mode; kept for compatibility)')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the RCCL gather of all frames to rank 0 after the timed region')
    ap.add_argument('--mode', default='shard', choices=['shard', 'clips', 'e2e'],
                    help='"shard" = ONE clip of N x frames sharded by frame range (Jacobi replicated, 16-byte crop all-reduce); '
                         '"clips" = N independent clips, one per GPU, no collective (BASELINE config 5); '
                         '"e2e" = host frames in -> host frames out, PCIe both ways (N independent clips when N > 1)')
    ap.add_argument('--no-e2e', action='store_true', help='skip the host-buffers-in / host-buffers-out side measurement (N = 1 only)')
    ap.add_argument('--no-workloads', action='store_true', help='skip the other configurations measured in the same run (N = 1, default workload only: '
                    'cfg3 and cfg4shard, ~4 s each, ~4 GB of HBM and of host memory)')
    ap.add_argument('--workload-steps', type=int, default=10, help='timed steps of each of those other configurations')
    ap.add_argument('--pipeline', default='product', choices=['product', 'serial'],
                    help='"product" = MeshFlowStabilizer.stabilize_resident\'s own overlap (sweep, tables, crop scan and rectangle on its prep '
                         'stream beside the warp chunks); "serial" = every kernel of a clip in order on one stream')
    ap.add_argument('--rectangle', default='', choices=['', 'fused', 'early'], help='MeshFlowStabilizer.resident_rectangle for this run')
    ap.add_argument('--chunks', type=int, default=-1, help='MeshFlowStabilizer.resident_chunks for this run (-1 = the product default, 0: in order, '
                    'the warp alone; k >= 1: k frame ranges with tables, crop scan and rectangle beside the warp)')
    ap.add_argument('--checksum', action='store_true', help='add `frames_checksum` to the line: one position-weighted 63-bit sum per stabilized '
                    'frame of the last step (the gathered clip on rank 0 when the gather ran, else this rank\'s shard) -- for the tests')
    ap.add_argument('--as-rank-of', type=int, default=0, metavar='N',
                    help='single process only: do the work rank 0 of an N-GPU "shard" run does (clip of N x frames, own '
                         'frame range, replicated Jacobi) without the collective -- predicts weak scaling on one GPU')
    return ap


def main():
    args = build_parser().parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_children(args))

    import torch
    import torch.distributed as dist
    from meshflow_amd import dist as mfdist, host, ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer

    rank, world, device = mfdist.init_from_env('cuda')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    H, W, per_gpu, R, C, omega, iters = WORKLOADS[args.workload]
    sliced = args.frames > 0 and args.frames != per_gpu
    if args.frames > 0:
        per_gpu = args.frames
    e2e_mode = args.mode == 'e2e'
    clips_mode = (args.mode == 'clips' and world > 1) or e2e_mode
    if clips_mode:                       # every rank owns a whole clip of its own (seed = rank)
        F, shard = per_gpu, (1, 0)
    elif args.as_rank_of > 1 and world == 1:
        F, shard = per_gpu * args.as_rank_of, (args.as_rank_of, 0)
    else:
        F, shard = per_gpu * world, (world, rank)
    lo, hi = host.shard_range(F, *shard)

    stab = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega,
                              optimization_num_iterations=iters, device=str(device))
    seed = rank if clips_mode else 0
    disp, hom = synthetic.motion(F, R, C, seed=seed)
    d_disp = torch.from_numpy(disp).to(device)
    d_frames = synthetic.frames_torch(hi - lo, H, W, device, seed=seed, kind=args.frames_kind, first_frame=lo)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        return float(mfdist.all_reduce_max(t).item())

    if world > 1:                                    # pre-flight, to stderr: what a first multi-GPU run needs to be debugged from its log
        try:
            free_b, total_b = torch.cuda.mem_get_info(device)
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == 'nccl' else 'n/a'
            print(f'[rank {rank}/{world}] {torch.cuda.get_device_name(device)} {device}: {free_b / 2**30:.1f} of {total_b / 2**30:.1f} GiB free; '
                  f'backend {dist.get_backend()} (RCCL {ver}); HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}', file=sys.stderr)
        except Exception as e:                       # never let the report take the run down
            print(f'[rank {rank}] pre-flight report failed: {e}', file=sys.stderr)
    comm = {'world_size': dist.get_world_size() if world > 1 else 1,
            'backend': (dist.get_backend() if world > 1 else None),
            'note': 'world size as the torch.distributed communicator reports it ("nccl" is RCCL on ROCm)'}

    if e2e_mode:
        # host frames in -> host frames out; K timed clips after W untimed ones, barrier on both sides
        frames_h = [f.copy() for f in d_frames.cpu().numpy()]
        del d_frames
        host_clip(stab, frames_h, disp, hom, max(args.warmup, 1))
        barrier()
        t0 = time.perf_counter()
        times = host_clip(stab, frames_h, disp, hom, args.steps)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({
                'metric': 'frames/sec end-to-end stabilize_clip(): host frames in -> stabilized host frames + crop bounds + paths + score out (PCIe both ways)',
                'value': F * world * args.steps / elapsed, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 1),
                'ms_per_step': elapsed / args.steps * 1e3, 'min_ms_per_step_rank0': float(np.min(times)) * 1e3,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
                'data': f'synthetic ({args.frames_kind} frames, injected random mesh motion, seed = rank)',
                'config': {'workload': f'{args.workload}: {W}x{H}, {per_gpu} frames per clip, {R}x{C} mesh, omega={omega}, {iters} Jacobi sweeps, '
                                       f'ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL; pageable NumPy frame list in, list out',
                           'parallelism': f'{world} independent clip(s), one per GPU, each over its own PCIe link'},
                'roofline': None, 'cpu_baseline': None, 'communicator': comm,
                'note': 'PCIe-inclusive rate; the HBM-resident rate and the kernel roofline are the default mode\'s line'}))
        if world > 1:
            dist.destroy_process_group()
        return

    no_events = bool(os.environ.get('MF_BENCH_NO_EVENTS'))     # tuning aid: what the HIP events around the kernels cost the step
    main_stream = torch.cuda.current_stream(device)
    if os.environ.get('MF_MAIN_PRIORITY'):                      # tuning aid: the warp's stream at another priority than the prep stream
        torch.cuda.synchronize()
        main_stream = torch.cuda.Stream(device=device, priority=int(os.environ['MF_MAIN_PRIORITY']))
        torch.cuda.set_stream(main_stream)
    if args.chunks >= 0:
        stab.resident_chunks = args.chunks
    if args.rectangle:
        stab.resident_rectangle = args.rectangle   # 'early': the rectangle from the table on the prep stream (the all-reduce of a sharded clip
                                                   # then runs beside the warp); the default takes it from the warp's own scan, behind the warp
    if os.environ.get('MF_RESIDENT_GATE'):                       # tuning aid
        stab.resident_gate = os.environ['MF_RESIDENT_GATE']
    serial_mode = args.pipeline == 'serial'
    kp = KernelPath(stab, d_frames, d_disp, hom, F, (lo, hi), W, H, R, C, device, collective=(world > 1 and not clips_mode), no_events=no_events)
    m = kp.measure(args.steps, args.warmup, serial_mode, barrier, max_over_ranks)
    elapsed, spinup_steps, bounds, d_out = m['elapsed'], m['spinup_steps'], m['bounds'], kp.d_out
    ev, jev = m['ev'], m['jev']

    # Beside the timed figure, outside the timed region (N = 1): the latency of ONE clip through the same pipeline from an idle GPU
    # (synchronise, issue one clip, synchronise: the sweep and the first frame range's table cannot hide behind anything), and the
    # K steps again with everything on one stream.
    extra = {}
    if world == 1 and not serial_mode and not os.environ.get('MF_BENCH_NO_EXTRAS'):
        extra['latency_ms_single_clip'] = kp.latency()
        extra['serial'] = kp.serial(args.steps, max(args.warmup, 3))

    warp_ms = float('nan') if no_events else float(np.mean([a.elapsed_time(b) for a, b in ev]))
    warp_alone = kp.warp_alone_ms(m['d_stab']) if (not serial_mode and stab._sweep_gate(d_disp) == 'plan') else None
    if os.environ.get('MF_BENCH_PER_STEP') and rank == 0 and not no_events:          # tuning aid: the launches one by one (clock transients)
        print('warp ms per step:', [round(a.elapsed_time(b), 3) for a, b in ev], file=sys.stderr)
    jac_ms = float('nan') if no_events else float(np.mean([a.elapsed_time(b) for a, b in jev]))
    # Jacobi kernel alone (the per-step figure above includes the host-side coefficient set-up), outside the timed region
    jac_kernel_ms, jac_flops, jac_series = kp.jacobi_kernel_ms(omega, iters)
    # Next row on the path, measured outside the timed region: crop to the clip-level bounds and
    # resize back, device-resident (mfs.py:1111-1157).
    resize_ms = None
    if not (bounds[2] < bounds[0] or bounds[3] < bounds[1]):
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch = torch.empty_like(d_frames)
        rect = bounds.tolist()                       # one D2H of the 4 bounds, outside the timed launches
        for _ in range(30):                          # first touches of the fresh output stack + the clock transient after the
            ops.crop_resize(d_out, rect, out=scratch)    # idle milliseconds of the allocation above (launches 3-20 run up to 40 % slow)
        r0.record()
        for _ in range(20):
            ops.crop_resize(d_out, rect, out=scratch)
        r1.record()
        torch.cuda.synchronize()
        resize_ms = r0.elapsed_time(r1) / 20
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
            motion_row = {'kernels': 'feature_prep + bitonic sort + vertex_row_median + median_blur + accumulate',
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
    # draining its own shard over its own PCIe link).  Both figures are reported side by side;
    # --no-gather skips them.
    gather_ms = gather_error = d2h_ms = checksum = None
    if args.checksum and not (world > 1 and not clips_mode and not args.no_gather):
        checksum = frame_checksums(d_out)
    if world > 1 and not clips_mode and not args.no_gather:
        try:
            # pre-flight: rank 0 receives G x ceil(F/G) frames next to its own shard, input and output; every rank agrees on whether it fits
            per_rank = -(-F // world)
            need = per_rank * H * W * 3 * (world if rank == 0 else 0) + (per_rank - (hi - lo)) * H * W * 3
            free_b, _ = torch.cuda.mem_get_info(device)
            fits = torch.tensor([1.0 if need < 0.9 * free_b else 0.0], dtype=torch.float64, device=device)
            fits = -mfdist.all_reduce_max(-fits)                     # min over ranks
            if rank == 0:
                print(f'[rank 0] gather needs {need / 2**30:.1f} GiB on rank 0, {free_b / 2**30:.1f} GiB free', file=sys.stderr)
            if float(fits.item()) < 0.5:
                raise MemoryError(f'gather skipped: rank 0 needs {need / 2**30:.1f} GiB, {free_b / 2**30:.1f} GiB free')
            barrier()
            t1 = time.perf_counter()
            gathered = mfdist.gather_frames(d_out, F)
            barrier()
            gather_ms = max_over_ranks(time.perf_counter() - t1) * 1e3
            if args.checksum and rank == 0:
                checksum = frame_checksums(gathered)
            del gathered
            pinned = torch.empty(d_out.shape, dtype=torch.uint8, pin_memory=True)
            pinned.copy_(d_out)                      # touch the pages once
            barrier()
            t1 = time.perf_counter()
            pinned.copy_(d_out, non_blocking=True)
            barrier()
            d2h_ms = max_over_ranks(time.perf_counter() - t1) * 1e3
            del pinned
        except Exception as e:                      # never let the optional collective take the measurement down
            gather_error = f'{type(e).__name__}: {e}'

    if rank == 0:
        algo_bytes = 2.0 * H * W * 3 * (hi - lo)
        achieved = algo_bytes / (warp_ms * 1e-3)
        traffic = None            # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload
        try:
            with open(os.path.join(REPO, 'profiles', 'traffic.json')) as fh:
                traffic = None if sliced else json.load(fh).get(args.workload, {}).get('traffic_bytes')
        except (OSError, ValueError):
            pass
        result = {
            'metric': 'frames/sec stabilize() hot path (Jacobi + mesh warp + crop scan), inputs resident in HBM -- the kernel-path figure of '
                      'BASELINE.json\'s metric; its host-to-host figure (PCIe both ways) is `end_to_end`, the warp kernel\'s %HBM-roofline is `roofline`',
            'value': (F * world if clips_mode else F) * args.steps / elapsed, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'spinup_steps': spinup_steps, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'pipeline': ('serial: every kernel of a clip in order on one stream' if serial_mode else
                         ('product (MeshFlowStabilizer.stabilize_resident), in order: cell table + plan -> warp alone -> rectangle on the main stream, the Jacobi '
                          'sweep on the prep stream (the NEXT clip\'s sweep runs beside THIS clip\'s table + plan, never beside a warp)' if stab.resident_chunks <= 0 else
                          f'product (MeshFlowStabilizer.stabilize_resident), {stab.resident_chunks} frame ranges: sweep + tables + crop scan + rectangle on the prep '
                          'stream beside the warp chunks on the main stream') +
                         '; clips issued back to back, `spinup_steps` untimed steps in front of the warm-up; `latency_ms_single_clip` and `serial` beside it'),
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'dtype_note': 'float64 vertex paths and pixel coordinates (as the reference), integer fixed-point interpolation on uint8',
            'data': f'synthetic ({args.frames_kind} frames, injected random mesh motion, seed 0)',
            'config': {'workload': f'{args.workload}{" (SLICE: --frames)" if sliced else ""}: {W}x{H}, {per_gpu} frames/GPU ({per_gpu * world} total), {R}x{C} mesh, '
                                   f'omega={omega}, {iters} Jacobi sweeps, ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL',
                       'parallelism': (f'{world} independent clips, one per GPU, no collective' if clips_mode else
                                       f'frame-range shards x{world}, Jacobi replicated, 16-byte crop all-reduce')},
            'roofline': {'kernel': 'warp_kernel', 'bound': 'hbm', 'achieved': achieved / 1e9,
                         'peak': HBM_PEAK_BYTES_PER_S / 1e9, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_BYTES_PER_S,
                         **({} if sliced else hbm_roofline_extras(algo_bytes, warp_ms, args.workload)),
                         'traffic': traffic, 'traffic_source': 'profiles/traffic.json (PMC: size-resolved TCC_EA0_RDREQ read requests + WRITE_SIZE)'
                         if traffic else None,
                         'algorithmic_bytes_per_launch': algo_bytes, 'avg_launch_ms': warp_ms,
                         'beside_the_warp': 'nothing' if (serial_mode or stab._sweep_gate(d_disp) != 'plan') else 'the next clip\'s Jacobi sweep (prep stream, gate "plan")',
                         'launches': (1 if serial_mode or stab.resident_chunks <= 0 else min(stab.resident_chunks, hi - lo)),
                         'launch_note': ('HIP events on the stream the warp kernel is launched on, directly in front of and behind it (nothing runs beside it in this '
                                         'arrangement)' if (serial_mode or stab.resident_chunks <= 0) else
                                         'one "launch" = the warp of ALL the step\'s frames: HIP events on the main stream in front of the first and behind the '
                                         'last of its `launches` warp kernels (frame ranges of the clip), so the figure includes the gaps between them and '
                                         'the prep-stream kernels (next clip\'s sweep, tables, crop scan) sharing the chip; the kernel trace under profiles/ '
                                         'gives the kernels alone'),
                         'note': 'bound by vector and scalar instruction issue (float64 coordinates, integer blend, one wavefront per 32x8 footprint), not by HBM: DESIGN.md 4.3',
                         **({'alone': {'avg_launch_ms': warp_alone, 'frac': algo_bytes / (warp_alone * 1e-3) / HBM_PEAK_BYTES_PER_S,
                                       'frac_of_achievable': algo_bytes / (warp_alone * 1e-3) / HBM_ACHIEVABLE_BYTES_PER_S,
                                       'note': 'the same launch with nothing beside it, 8 launches between two events outside the timed steps'}} if warp_alone else {})},
            'jacobi': {'avg_ms_in_pipeline': jac_ms, 'on_prep_stream': not serial_mode, 'kernel_ms': jac_kernel_ms,
                       'note': 'avg_ms_in_pipeline: HIP events around the stage (coefficient upload + sweep) on the stream it is issued on -- on the prep '
                               'stream it shares the chip with the previous clip\'s warp, so this is NOT the kernel\'s own time; kernel_ms: the kernel alone, measured after the timed region', 'series': jac_series, 'frames': F,
                       'bound': 'fp64 vector ALU + LDS (the state never leaves the chip)', 'achieved': jac_flops / (jac_kernel_ms * 1e-3) / 1e12,
                       'peak': 78.6, 'unit': 'TFLOP/s', 'frac': jac_flops / (jac_kernel_ms * 1e-3) / 78.6e12},
            'crop_bounds': [int(v) for v in bounds.tolist()],
            'communicator': comm,
            'cfg1': cfg1_status(),
        }
        result.update(extra)
        # the same figures under one key (the bench contract of this tier keeps the HBM-resident rate in `value`: "inputs already resident in
        # HBM when the timed region starts ... the PCIe-inclusive rate is never `value`"; BASELINE.json's host-to-host figure is `end_to_end`,
        # a timed region of its own with its own `steps` / `ms_per_step` / `elapsed_s`)
        result['kernel_path'] = {'value': result['value'], 'unit': 'frames/s', 'ms_per_step': result['ms_per_step'], 'steps': args.steps,
                                 'through': 'MeshFlowStabilizer.stabilize_resident (public method, deferred degenerate-mesh check, finish() inside the timed region)'
                                 if not serial_mode else 'private stages in order on one stream',
                                 'serial': extra.get('serial'), 'latency_ms_single_clip': extra.get('latency_ms_single_clip')}
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
            result['sharded_d2h_ms'] = d2h_ms
            result['gather_note'] = ('after the timed region, max over ranks: gather_to_rank0_ms = one gather of every rank\'s stabilized frames '
                                     'to rank 0 over xGMI (north_star\'s final gather); sharded_d2h_ms = every rank draining its own shard to '
                                     'pinned host memory over its own PCIe link (what a host-memory consumer would use instead)')
        if gather_error is not None:
            result['gather_error'] = gather_error
        if checksum is not None:
            result['frames_checksum'] = checksum
            result['frames_checksum_range'] = [0, F] if gather_ms is not None else [lo, hi]
        if world == 1 and not args.no_e2e and args.as_rank_of <= 1:
            try:
                result['end_to_end'] = end_to_end(stab, d_frames, disp, hom, F)
                # north_star's ">= 500 frames/sec end-to-end stabilize()" is quoted on THIS figure (host frames in -> cropped host frames
                # out, PCIe both ways); `value` stays the HBM-resident rate the bench contract asks for
                result['config']['end_to_end_fps'] = result['end_to_end']['with_crop']['value']
                result['config']['end_to_end_fps_note'] = ('stabilize_clip(crop=True): host frames in -> stabilized, cropped + resized host frames out, '
                                                           'PCIe both ways, same run; detail under `end_to_end`; north_star target: 500')
                result['roofline']['end_to_end_fps'] = result['end_to_end']['with_crop']['value']
            except Exception as e:                      # (host memory): never let the side measurement take the line down
                result['end_to_end'] = {'error': f'{type(e).__name__}: {e}'}
        if world == 1 and args.workload == 'cfg2' and not sliced and not args.no_workloads and args.as_rank_of <= 1 and not serial_mode:
            # the other single-GPU configurations in the same run, so that whoever runs this line observes them too
            del kp, d_out, d_frames
            torch.cuda.empty_cache()
            pcie = (result.get('end_to_end') or {}).get('pcie_probe') or {}
            result['workloads'] = {}
            for name in ('cfg3', 'cfg4shard'):
                try:
                    result['workloads'][name] = other_workload(name, device, args.workload_steps, pcie)
                except Exception as e:
                    result['workloads'][name] = {'error': f'{type(e).__name__}: {e}'}
                torch.cuda.empty_cache()
        if world == 1 and args.cpu_frames > 0:
            try:
                result['cpu_baseline'], stab_cpu = cpu_baseline(H, W, F, R, C, omega, iters, disp, hom, args.cpu_frames)
            except Exception as e:                      # e.g. no C compiler and no prebuilt oracle on the box
                stab_cpu = None
                result['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': 0, 'kind': 'port',
                                          'sample': f'failed: {type(e).__name__}: {e}'}
            if stab_cpu is not None and not args.no_faithful:
                try:
                    result['cpu_baseline']['reference_faithful'] = cpu_reference_faithful(H, W, F, R, C, omega, iters, disp, hom, stab_cpu)
                except Exception as e:
                    result['cpu_baseline']['reference_faithful'] = {'value': None, 'kind': 'reference-faithful', 'sample': f'failed: {type(e).__name__}: {e}'}
        else:
            result['cpu_baseline'] = None
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""ONE process drives the N GPUs of a node through the C ABI alone (no torch.distributed): the frame-range sharded hot path with the
exchange steps of csrc/comm.hip -- mf_comm_init_all (ncclCommInitAll), mf_allreduce_crop, mf_gather_frames.

    python tools/capi_shard_run.py [--gpus N] [--workload small|cfg2] [--frames-per-gpu K]

Per device g (a host thread each): its frame range of the clip up, the replicated Jacobi sweep (mf_jacobi_f64), cell table + crop scan +
rectangle + warp (mf_warp_clip_u8c3).  Then the 16-byte all-reduce of the rectangles and the gather of every shard to device 0, and a
check: the gathered clip and the rectangle must equal ONE device doing the whole clip (done on device 0 when it fits).  Prints one JSON
line.  With N = 1 (the build's GPU boxes) the communicator has one rank: what the N > 1 run adds is RCCL itself."""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=0)
    ap.add_argument('--workload', default='cfg2', choices=['small', 'cfg2'])
    ap.add_argument('--frames-per-gpu', type=int, default=0)
    args = ap.parse_args()
    import bench
    from meshflow_amd import _lib, host, synthetic
    lib = _lib.lib
    count = ctypes.c_int(0)
    _lib.check(lib.mf_device_count(ctypes.byref(count)))
    G = args.gpus if args.gpus > 0 else count.value
    G = min(G, count.value)
    H, W, per, R, C, omega, iters = bench.WORKLOADS[args.workload]
    if args.frames_per_gpu:
        per = args.frames_per_gpu
    per = min(per, 64) if args.workload == 'cfg2' and not args.frames_per_gpu else per      # bounded host memory for the reference run
    F = per * G
    disp, hom = synthetic.motion(F, R, C, seed=0)
    taps, lam, inv_on = host.jacobi_band_coefficients(F, W, H, 0, hom, omega)
    S = (R + 1) * (C + 1) * 2
    fb = H * W * 3
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    border = (ctypes.c_uint8 * 3)(0, 0, 255)

    class Dev:
        pass

    devs = [Dev() for _ in range(G)]
    errors = []

    def dmalloc(nbytes):
        ptr = ctypes.c_void_p()
        _lib.check(lib.mf_malloc(ctypes.byref(ptr), nbytes))
        return ptr

    def work(g):
        try:
            d = devs[g]
            _lib.check(lib.mf_set_device(g))
            lo, hi = host.shard_range(F, G, g)
            d.lo, d.hi, n = lo, hi, hi - lo
            frames = synthetic.frames_numpy(n, H, W, seed=0, kind='pattern', first_frame=lo)
            d.frames = dmalloc(n * fb); d.out = dmalloc(n * fb)
            d.b = dmalloc(F * S * 8); d.x = dmalloc(F * S * 8)
            d.taps = dmalloc(taps.nbytes); d.lam = dmalloc(lam.nbytes); d.inv = dmalloc(inv_on.nbytes)
            d.table = dmalloc(lib.mf_cell_table_bytes(n, W, H, R, C))
            d.crop = dmalloc(n * 16); d.bounds = dmalloc(16); d.status = dmalloc(4)
            zero = np.zeros(1, np.int32)
            for dst, src in ((d.frames, frames), (d.b, np.ascontiguousarray(disp)), (d.taps, taps), (d.lam, lam), (d.inv, inv_on), (d.status, zero)):
                _lib.check(lib.mf_memcpy_h2d(dst, p(src), src.nbytes, None))
            _lib.check(lib.mf_stream_synchronize(None))
            t0 = time.perf_counter()
            _lib.check(lib.mf_jacobi_f64(d.b, d.x, d.taps, d.lam, d.inv, F, S, omega, iters, None))
            off = lo * S * 8
            _lib.check(lib.mf_warp_clip_u8c3(d.frames, d.out, ctypes.c_void_p(d.b.value + off), ctypes.c_void_p(d.x.value + off), n, W, H, R, C,
                                             border, d.table, d.crop, d.bounds, d.status, 4, None, None))
            _lib.check(lib.mf_stream_synchronize(None))
            d.compute_s = time.perf_counter() - t0
            st = np.zeros(1, np.int32)
            _lib.check(lib.mf_memcpy_d2h(p(st), d.status, 4, None)); _lib.check(lib.mf_stream_synchronize(None))
            if st[0]:
                raise ValueError(f'{st[0]} degenerate cells on device {g}')
        except Exception as e:                 # noqa: BLE001
            errors.append(f'device {g}: {type(e).__name__}: {e}')

    threads = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    t_all = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print(json.dumps({'tool': 'capi_shard_run', 'failed': errors}))
        return 1
    compute_wall = time.perf_counter() - t_all
    _lib.check(lib.mf_comm_init_all(G))
    try:
        t0 = time.perf_counter()
        _lib.check(lib.mf_allreduce_crop((ctypes.c_void_p * G)(*[d.bounds.value for d in devs])))
        allreduce_ms = (time.perf_counter() - t0) * 1e3
        _lib.check(lib.mf_set_device(0))
        dst = dmalloc(F * fb)
        sizes = (ctypes.c_size_t * G)(*[(d.hi - d.lo) * fb for d in devs])
        t0 = time.perf_counter()
        _lib.check(lib.mf_gather_frames((ctypes.c_void_p * G)(*[d.out.value for d in devs]), sizes, dst, 0))
        gather_ms = (time.perf_counter() - t0) * 1e3
        rects = []
        for g, d in enumerate(devs):
            _lib.check(lib.mf_set_device(g))
            r = np.zeros(4, np.int32)
            _lib.check(lib.mf_memcpy_d2h(p(r), d.bounds, 16, None)); _lib.check(lib.mf_stream_synchronize(None))
            rects.append(r.tolist())
        _lib.check(lib.mf_set_device(0))
        got = np.empty((F, H, W, 3), np.uint8)
        _lib.check(lib.mf_memcpy_d2h(p(got), dst, F * fb, None)); _lib.check(lib.mf_stream_synchronize(None))
        # reference: the whole clip on device 0 through the host-buffer entry point
        frames = synthetic.frames_numpy(F, H, W, seed=0, kind='pattern')
        x = np.empty((F, S))
        _lib.check(lib.mf_jacobi_f64_host(p(np.ascontiguousarray(disp.reshape(F, S))), p(x), p(taps), p(lam), p(inv_on), F, S, omega, iters, None))
        want = np.empty_like(frames)
        crop = np.zeros((F, 4), np.int32)
        _lib.check(lib.mf_warp_u8c3_host(p(frames), p(want), p(np.ascontiguousarray(disp)), p(x), F, W, H, R, C, border, p(crop), None))
        rect = [int(crop[:, 0].max()), int(crop[:, 1].max()), int(crop[:, 2].min()), int(crop[:, 3].min())]
        ok = bool(np.array_equal(got, want)) and all(r == rect for r in rects)
        print(json.dumps({'tool': 'capi_shard_run', 'n_gpus': G, 'workload': f'{args.workload}: {W}x{H}, {per} frames per GPU ({F} total)',
                          'gathered_clip_and_rectangle_equal_one_device': ok, 'rectangle': rect, 'per_device_rectangles_after_allreduce': rects,
                          'allreduce_crop_ms': allreduce_ms, 'gather_frames_ms': gather_ms, 'gather_GBps': F * fb / gather_ms / 1e6,
                          'compute_wall_ms_incl_upload': compute_wall * 1e3, 'compute_ms_per_device': [d.compute_s * 1e3 for d in devs]}))
        return 0 if ok else 1
    finally:
        lib.mf_comm_destroy()


if __name__ == '__main__':
    sys.exit(main())

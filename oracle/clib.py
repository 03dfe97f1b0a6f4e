"""ctypes loader for the C oracle (oracle/warp_oracle.c, oracle/motion_oracle.c).  TEST INFRASTRUCTURE ONLY.

`load()` builds `oracle/_build/liboracle*.so` with `make -C oracle` when missing (gcc only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

CELL_DOUBLES = 32
OFF_M, OFF_HI, OFF_RECT, OFF_BBOX, OFF_STATUS = 0, 9, 18, 22, 26

_dp = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_fp = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.run(['make', '-C', _HERE], check=True, stdout=subprocess.DEVNULL)


def load(openmp=False):
    name = 'liboracle_omp.so' if openmp else 'liboracle.so'
    if name in _LIBS:
        return _LIBS[name]
    path = os.path.join(_HERE, '_build', name)
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ('warp_oracle.c', 'motion_oracle.c'))
    if not os.path.exists(path) or os.path.getmtime(path) < newest:
        build()
    lib = ctypes.CDLL(path)
    lib.mfo_jacobi_banded.argtypes = [_dp, _dp, _dp, _dp, _dp] + [ctypes.c_int] * 4
    lib.mfo_jacobi_banded.restype = None
    lib.mfo_find_homography_4pt.argtypes = [_dp, _dp, _dp]
    lib.mfo_find_homography_4pt.restype = ctypes.c_int
    lib.mfo_invert3x3.argtypes = [_dp, _dp]
    lib.mfo_invert3x3.restype = None
    lib.mfo_cell_table.argtypes = [ctypes.c_int] * 4 + [_dp, _dp, _dp]
    lib.mfo_cell_table.restype = ctypes.c_int
    lib.mfo_warp_frame.argtypes = [_u8p, _u8p] + [ctypes.c_int] * 4 + [_dp, _u8p, ctypes.c_int, _i32p, _fp, _fp]
    lib.mfo_warp_frame.restype = None
    lib.mfo_warp_clip.argtypes = [_u8p, _u8p] + [ctypes.c_int] * 5 + [_dp, _dp, _u8p, ctypes.c_int, _i32p]
    lib.mfo_warp_clip.restype = ctypes.c_int
    lib.mfo_set_threads.argtypes = [ctypes.c_int]
    lib.mfo_set_threads.restype = ctypes.c_int
    lib.mfo_cell_doubles.restype = ctypes.c_int
    lib.mfo_vertex_motion.argtypes = [_dp, _dp, _i32p, _dp] + [ctypes.c_int] * 7 + [_fp, _dp]
    lib.mfo_vertex_motion.restype = ctypes.c_int
    assert lib.mfo_cell_doubles() == CELL_DOUBLES
    if openmp:
        # OpenMP's default is one thread per CPU the HOST shows (hundreds on the GPU boxes, whose containers may use 16): cap it to what
        # this process may really use -- affinity mask and cgroup quota --, at most 32.  (bench.py's cpu_baseline sets its own count.)
        lib.mfo_set_threads(max(1, min(_usable_cpus(), 32)))
    _LIBS[name] = lib
    return lib


def _usable_cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (('/sys/fs/cgroup/cpu.max', lambda t: t.split()), ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', lambda t: [t.strip(), None])):
        try:
            with open(path) as fh:
                quota, period = parse(fh.read())
            if period is None:
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh:
                    period = fh.read().strip()
            if quota not in ('max', '-1'):
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return n


def _p(a, t):
    return a.ctypes.data_as(t)


def jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=False):
    b = np.ascontiguousarray(b, dtype=np.float64)
    F, S = b.shape
    x = np.empty_like(b)
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    inv_on = np.ascontiguousarray(inv_on, dtype=np.float64)
    load(openmp).mfo_jacobi_banded(_p(b, _dp), _p(x, _dp), _p(taps, _dp), _p(lam, _dp), _p(inv_on, _dp), F, S, omega, iters)
    return x


def find_homography_4pt(src, dst):
    s = np.ascontiguousarray(src, dtype=np.float64).reshape(8)
    d = np.ascontiguousarray(dst, dtype=np.float64).reshape(8)
    H = np.empty(9)
    ok = load().mfo_find_homography_4pt(_p(s, _dp), _p(d, _dp), _p(H, _dp))
    return H.reshape(3, 3) if ok else None


def invert3x3(S):
    s = np.ascontiguousarray(S, dtype=np.float64).reshape(9)
    t = np.empty(9)
    load().mfo_invert3x3(_p(s, _dp), _p(t, _dp))
    return t.reshape(3, 3)


def cell_table(W, H, R, C, unstab_f, stab_f):
    u = np.ascontiguousarray(unstab_f, dtype=np.float64).reshape(-1)
    s = np.ascontiguousarray(stab_f, dtype=np.float64).reshape(-1)
    assert u.size == (R + 1) * (C + 1) * 2 == s.size
    table = np.zeros((R * C, CELL_DOUBLES))
    bad = load().mfo_cell_table(W, H, R, C, _p(u, _dp), _p(s, _dp), _p(table, _dp))
    return table, bad


def warp_frame(frame, R, C, table, border_bgr=(0, 0, 255), use_bbox=False, want_maps=False):
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    H, W = frame.shape[:2]
    out = np.empty_like(frame)
    table = np.ascontiguousarray(table, dtype=np.float64)
    border = np.asarray(border_bgr, dtype=np.uint8)
    crop = np.zeros(4, dtype=np.int32)
    mx = np.empty((H, W), dtype=np.float32) if want_maps else None
    my = np.empty((H, W), dtype=np.float32) if want_maps else None
    load().mfo_warp_frame(_p(frame, _u8p), _p(out, _u8p), W, H, R, C, _p(table, _dp), _p(border, _u8p),
                          int(use_bbox), _p(crop, _i32p),
                          _p(mx, _fp) if want_maps else None, _p(my, _fp) if want_maps else None)
    if want_maps:
        return out, crop, mx, my
    return out, crop


def warp_clip(frames, R, C, unstab, stab, border_bgr=(0, 0, 255), use_bbox=False, openmp=False):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, H, W = frames.shape[:3]
    u = np.ascontiguousarray(unstab, dtype=np.float64)
    s = np.ascontiguousarray(stab, dtype=np.float64)
    out = np.empty_like(frames)
    crop = np.zeros((n, 4), dtype=np.int32)
    border = np.asarray(border_bgr, dtype=np.uint8)
    bad = load(openmp).mfo_warp_clip(_p(frames, _u8p), _p(out, _u8p), n, W, H, R, C, _p(u, _dp), _p(s, _dp),
                                    _p(border, _u8p), int(use_bbox), _p(crop, _i32p))
    return out, crop, bad


def set_threads(n):
    """Thread count of the OpenMP build; returns the count in effect."""
    return load(True).mfo_set_threads(int(n))


def pack_features(features_by_pair):
    """[(early (K,1,2), late (K,1,2)) or (None, None), ...] -> (early (Ktot,2) f64, late (Ktot,2) f64, offsets int32)."""
    early = [np.zeros((0, 2)) if e is None else np.asarray(e, dtype=np.float64).reshape(-1, 2) for e, _ in features_by_pair]
    late = [np.zeros((0, 2)) if l is None else np.asarray(l, dtype=np.float64).reshape(-1, 2) for _, l in features_by_pair]
    offsets = np.cumsum([0] + [len(e) for e in early]).astype(np.int32)
    cat = lambda parts: np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0, 2))
    return cat(early), cat(late), offsets


def vertex_motion(W, H, R, C, ell_rows, ell_cols, features_by_pair, homographies, openmp=False):
    """(displacements float64 (P+1,R+1,C+1,2), velocities float32 (P,R+1,C+1,2)); ValueError like math.sqrt."""
    early, late, offsets = pack_features(features_by_pair)
    P = len(features_by_pair)
    hom = np.ascontiguousarray(np.asarray(homographies, dtype=np.float64)[:P]).reshape(P, 9)
    vel = np.zeros((P, R + 1, C + 1, 2), dtype=np.float32)
    disp = np.zeros((P + 1, R + 1, C + 1, 2), dtype=np.float64)
    status = load(openmp).mfo_vertex_motion(_p(early, _dp), _p(late, _dp), _p(offsets, _i32p), _p(hom, _dp), P,
                                            W, H, R, C, ell_rows, ell_cols, _p(vel, _fp), _p(disp, _dp))
    if status:
        raise ValueError('math domain error')
    return disp, vel

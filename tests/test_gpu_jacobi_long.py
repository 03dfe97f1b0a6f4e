"""Clips of ANY length through mf_jacobi_f64 (mfs.py:193-213 reads every frame of the file; mfs.py:632-710, 871-878 work for any
num_frames): beyond the 9,728 frames the LDS kernels hold, the sweep runs in time tiles with a halo of (sweeps per launch) x omega
frames (csrc/jacobi.hip, launch_jacobi_tiled), and beyond a radius of 246 sweep by sweep in global memory.  Everything bit-identical
to the C oracle (taps ascending from zero, one fma each, then inv_on * fma(2 lam, acc, b)).  Needs a real MI355X (-m gpu)."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _problem(F, S, omega, seed=0, symmetric=True):
    from meshflow_amd import synthetic
    b = np.cumsum(2.0 * synthetic.normal(np.arange(F * S).reshape(F, S), seed=F + omega + seed), axis=0)
    taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
    if not symmetric:
        taps = taps * (1.0 + 0.01 * synthetic.uniform01(np.arange(2 * omega + 1), seed=11))
    lam = 0.95 * synthetic.uniform01(np.arange(F), seed=3)
    inv_on = 1.0 / (1 + 2 * lam * taps.sum())
    return b, taps, lam, inv_on


def _hip(dev, b, taps, lam, inv_on, omega, iters, reps=1):
    from meshflow_amd import ops
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    args = (t(b), t(taps), t(lam), t(inv_on), omega, iters)
    x = ops.jacobi(*args)
    torch.cuda.synchronize()
    ms = None
    if reps > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.jacobi(*args, out=x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    return x.cpu().numpy(), ms


class _Env:
    """The library's testing aids are read at every call (getenv): set for the duration of a `with` block."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize('F,S,omega,iters,tile', [
    (700, 5, 10, 100, 97),          # config 2's radius and sweeps: ONE launch, halo 1,000 > the whole clip, 8 seams
    (3000, 3, 10, 100, 1000),       # halo 1,000, tiles of 1,000
    (1500, 4, 30, 20, 333),         # config 3's radius (specialised tiled kernel), 20 sweeps
    (1500, 2, 30, 200, 500),        # 200 sweeps at omega = 30: three launches (67 + 67 + 66), scratch ping-pong
    (900, 3, 1, 7, 64), (900, 3, 7, 50, 200), (1200, 2, 13, 40, 301), (1200, 2, 25, 33, 400),      # run-time-radius tiled kernel
    (2000, 2, 40, 130, 700),        # ks_max = 60: three launches of the run-time-radius kernel
    (1000, 2, 100, 9, 400), (1500, 1, 246, 10, 800),                                                # the widest radius the tiles hold
    (5, 3, 10, 4, 2), (1, 2, 3, 5, 1), (64, 2, 10, 0, 16),                                          # tiny clips; no sweep at all
])
def test_tiled_sweep_bit_exact_on_forced_seams(dev, F, S, omega, iters, tile):
    """MF_JACOBI_LONG=1 sends a clip of any length through the time tiles, MF_JACOBI_TILE_T caps the frames a tile writes: seams every
    `tile` frames, halos that reach beyond both clip ends, several launches."""
    from oracle import clib
    b, taps, lam, inv_on = _problem(F, S, omega)
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
    with _Env(MF_JACOBI_LONG=1, MF_JACOBI_TILE_T=tile):
        got, _ = _hip(dev, b, taps, lam, inv_on, omega, iters)
    assert np.array_equal(got, want)
    with _Env(MF_JACOBI_LONG=1, MF_JACOBI_TILE_T=tile, MF_JACOBI_RUNTIME=1):            # the run-time-radius form of the same tiles
        got, _ = _hip(dev, b, taps, lam, inv_on, omega, iters)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('F,S,omega,iters', [(300, 70, 10, 9), (1000, 3, 30, 4), (257, 130, 300, 3), (40, 5, 3, 0), (1, 1, 1, 2)])
def test_global_memory_sweep_bit_exact(dev, F, S, omega, iters):
    """MF_JACOBI_LONG=2: one launch per sweep on the arrays in global memory (the form that takes radii beyond 246 on long clips)."""
    from oracle import clib
    b, taps, lam, inv_on = _problem(F, S, omega, symmetric=False)
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
    with _Env(MF_JACOBI_LONG=2):
        got, _ = _hip(dev, b, taps, lam, inv_on, omega, iters)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('F,omega,iters', [(12000, 10, 100), (20000, 10, 100), (12000, 30, 20), (20000, 30, 20)])
def test_long_clips_bit_exact_with_time(dev, F, omega, iters):
    """mf_jacobi_f64 == clib.jacobi_banded at F = 12,000 / 20,000 for (omega = 10, 100 sweeps) and
    (omega = 30, 20 sweeps), 16 x 16 mesh (578 series), with the time reported (gpurun_out/jacobi_long.jsonl)."""
    from oracle import clib
    S = 578
    b, taps, lam, inv_on = _problem(F, S, omega)
    t0 = time.perf_counter()
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
    cpu_s = time.perf_counter() - t0
    got, ms = _hip(dev, b, taps, lam, inv_on, omega, iters, reps=5)
    assert np.array_equal(got, want)
    flops = float(iters) * F * S * (2 * (2 * omega + 1) + 3)
    rec = {'F': F, 'S': S, 'omega': omega, 'iters': iters, 'ms': ms, 'tflops': flops / (ms * 1e-3) / 1e12, 'cpu_oracle_s': cpu_s}
    print('jacobi long clip:', json.dumps(rec))
    out = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'jacobi_long.jsonl'), 'a') as fh:
            fh.write(json.dumps(rec) + '\n')


@pytest.mark.parametrize('F,S,omega,iters', [(9729, 3, 10, 12), (10100, 2, 246, 3), (10100, 2, 247, 3), (12000, 4, 300, 2), (40000, 2, 5, 30)])
def test_long_clips_other_radii(dev, F, S, omega, iters):
    """The first frame count beyond the LDS kernels; the widest radius the tiles hold and the first one that goes sweep by sweep through
    global memory; a radius of hundreds of frames; a very long clip at a small radius (run-time-radius tiles)."""
    from oracle import clib
    b, taps, lam, inv_on = _problem(F, S, omega)
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
    got, _ = _hip(dev, b, taps, lam, inv_on, omega, iters)
    assert np.array_equal(got, want)


def test_stabilizer_accepts_a_long_clip(dev):
    """The drop-in method itself (mfs.py:632-710) on a clip beyond the old ceiling: 11,000 frames, 4 x 4 mesh."""
    import meshflow_amd as amd
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    F, R, C, W, H = 11000, 4, 4, 640, 360
    disp, hom = synthetic.motion(F, R, C, seed=5)
    s = amd.MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=30, device='cuda:0')
    frames0 = [np.zeros((H, W, 3), np.uint8)]
    got = s._get_stabilized_vertex_displacements(F, frames0, 0, disp, hom)
    from meshflow_amd import host
    from oracle import clib
    taps, lam, inv_on = host.jacobi_band_coefficients(F, W, H, 0, hom, 10)          # the product's O(F) coefficients: the sweep itself, bit for bit
    want = clib.jacobi_banded(disp.reshape(F, -1), taps, lam, inv_on, 10, 30, openmp=True).reshape(disp.shape)
    assert np.array_equal(got, want)
    taps, lam, on = mo.jacobi_band_coefficients(F, W, H, 0, hom, 10)                # the oracle's coefficients (row sums in another order)
    want = clib.jacobi_banded(disp.reshape(F, -1), taps, lam, np.reciprocal(on), 10, 30, openmp=True).reshape(disp.shape)
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


def test_jacobi_corner_shapes_equal_the_oracle():
    """The other end of the size range: 1,008 shapes with clips of 1-641 frames, radii from 1 to 300 (far beyond the clip), 0-7 sweeps,
    1-130 series -- every one bit-identical to the C oracle (mfs.py:871-876 works for any of them)."""
    import itertools
    import torch
    from meshflow_amd import ops
    from oracle import clib
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(3)
    for F, omega, iters, S in itertools.product((1, 2, 3, 5, 17, 64, 65, 321, 641), (1, 2, 10, 30, 33, 100, 300), (0, 1, 2, 7), (1, 2, 7, 130)):
        b = rng.normal(size=(F, S))
        taps = rng.uniform(0.1, 1.0, size=2 * omega + 1)
        lam = rng.uniform(0.0, 0.95, size=F)
        inv_on = 1.0 / (1.0 + 2.0 * rng.uniform(0.5, 3.0, size=F))
        want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters)
        got = ops.jacobi(torch.from_numpy(b).to(dev), torch.from_numpy(taps).to(dev), torch.from_numpy(lam).to(dev),
                         torch.from_numpy(inv_on).to(dev), omega, iters).cpu().numpy()
        assert np.array_equal(got, want), (F, omega, iters, S)


def test_seventy_thousand_frames_through_every_stage():
    """A clip of 70,000 frames (32 x 16 pixels: 39 minutes at 30 fps) -- more than one launch's 65,535 (the grid's y extent) -- through the
    device operators and through `stabilize_resident` / `stabilize_clip`: warp, scan, rectangle and crop + resize equal the oracles on
    frames at both ends and around frame 65,535; the paths within 1e-9 of the banded oracle.  (`mf_warp_u8c3` and `mf_crop_resize_u8c3`
    refused more than 65,535 frames per call until the end of round 5 although their launch loops handled them.)"""
    import torch
    from meshflow_amd import host, ops
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib, meshflow_oracle as mo
    dev = torch.device('cuda:0')
    F, H, W, R, C = 70000, 16, 32, 2, 3
    rng = np.random.default_rng(6)
    frames = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    vel = rng.normal(0, 0.05, size=(F, R + 1, C + 1, 2))
    vel[0] = 0
    disp = np.cumsum(vel, axis=0)
    hom = np.tile(np.eye(3), (F, 1, 1))
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=4, optimization_num_iterations=6, device='cuda:0')
    d_frames, d_disp = torch.from_numpy(frames).to(dev), torch.from_numpy(disp).to(dev)
    out, bounds, d_stab = s.stabilize_resident(d_frames, d_disp, hom)
    s.finish()
    stab = d_stab.cpu().numpy()
    taps, lam, inv_on = host.jacobi_band_coefficients(F, W, H, 0, hom, 4)
    want_stab = clib.jacobi_banded(disp.reshape(F, -1), taps, lam, inv_on, 4, 6).reshape(disp.shape)
    np.testing.assert_allclose(stab, want_stab, rtol=0, atol=1e-9)
    o = out.cpu().numpy()
    rows = []
    for lo, hi in ((0, 40), (65500, 65600), (69960, 70000)):
        want, want_crop, bad = clib.warp_clip(frames[lo:hi], R, C, disp[lo:hi], np.ascontiguousarray(stab[lo:hi]), (0, 0, 255))
        assert bad == 0 and np.array_equal(o[lo:hi], want), (lo, hi)
        rows.append(want_crop)
    table = ops.cell_table(d_disp, d_stab, W, H, R, C)
    ops.crop_scan(table)
    crop = table.crop.cpu().numpy()
    assert np.array_equal(crop[0:40], rows[0]) and np.array_equal(crop[65500:65600], rows[1]) and np.array_equal(crop[69960:], rows[2])
    rect = [int(crop[:, 0].max()), int(crop[:, 1].max()), int(crop[:, 2].min()), int(crop[:, 3].min())]
    assert bounds.tolist() == rect and ops.crop_reduce(table.crop, W, H).tolist() == rect
    resized = ops.crop_resize(out, rect).cpu().numpy()
    for lo, hi in ((0, 20), (65530, 65540), (69990, 70000)):
        assert np.array_equal(resized[lo:hi], np.stack(mo.crop_frames(list(o[lo:hi]), rect))), (lo, hi)
    host_out, host_rect, host_stab, score, cropped = s.stabilize_clip(list(frames), disp, hom, crop=True)
    assert [int(v) for v in host_rect] == rect and np.array_equal(host_stab, stab)
    for i in (0, 1, 65535, 65536, 69999):
        assert np.array_equal(host_out[i], o[i]) and np.array_equal(cropped[i], resized[i]), i

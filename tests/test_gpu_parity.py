"""Parity of the HIP path against the oracle, through the C ABI.  Needs a real MI355X (-m gpu).

Bars (BASELINE.json north_star): stabilized vertex paths within 1e-4 of the reference; warped uint8
pixels within 1 LSB.  What is asserted here is tighter: paths within 1e-9 of the reference's own
outputs and bit-identical to the C oracle; cell tables, pixels and crop values bit-identical to the
C oracle."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _coeffs(F, W, H, definition, hom, omega):
    from oracle import meshflow_oracle as mo
    taps, lam, on = mo.jacobi_band_coefficients(F, W, H, definition, hom, omega)
    return taps, lam, np.reciprocal(on)


def _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters):
    from meshflow_amd import ops
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    x = ops.jacobi(t(b), t(taps), t(lam), t(inv_on), omega, iters)
    torch.cuda.synchronize()
    return x.cpu().numpy()


# ---------------------------------------------------------------------------------------------- Jacobi

@pytest.mark.parametrize('F,S,omega', [(641, 6, 10), (1280, 5, 10), (2400, 7, 10), (2561, 3, 10), (5120, 2, 10), (9000, 2, 10),
                                       (700, 4, 30), (2400, 3, 30), (5000, 2, 30), (9728, 1, 30), (10100, 1, 30),
                                       (1300, 1100, 10),
                                       (300, 6, 5), (700, 3, 5), (2600, 2, 5), (300, 5, 15), (1100, 4, 15), (4000, 2, 15), (4200, 1, 15),
                                       (300, 5, 20), (2100, 3, 20), (4000, 2, 20),
                                       (300, 100, 10), (380, 300, 10), (500, 200, 10), (300, 578, 10), (300, 6, 40)])
def test_jacobi_long_clips_every_variant_vs_c_oracle(dev, F, S, omega):
    """Few series + long clips spread a series over 2-8 wavefronts (the replicated sweep of a multi-GPU run); many series
    keep one wavefront per series; beyond the specialised sizes the generic kernel takes over.  All bit-identical."""
    from meshflow_amd import synthetic
    from oracle import clib
    iters = 6
    b = np.cumsum(2.0 * synthetic.normal(np.arange(F * S).reshape(F, S), seed=F + omega), axis=0)
    taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
    lam = 0.95 * synthetic.uniform01(np.arange(F), seed=3)
    inv_on = 1.0 / (1 + 2 * lam * taps.sum())
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
    got = _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('definition', [0, 1, 2, 3])
def test_jacobi_small_vs_reference_golden(dev, golden_dir, definition):
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    g = np.load(os.path.join(golden_dir, 'jacobi_small.npz'))
    s = MeshFlowStabilizer(mesh_row_count=int(g['R']), mesh_col_count=int(g['C']),
                           temporal_smoothing_radius=int(g['omega']), optimization_num_iterations=int(g['iters']))
    frames = [np.zeros((int(g['height']), int(g['width']), 3), np.uint8)]
    got = s._get_stabilized_vertex_displacements(int(g['F']), frames, definition, g['disp'], g['hom'])
    want = g[f'stab_D{definition}']
    assert got.shape == want.shape and got.dtype == np.float64
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())      # bar: 1e-4


@pytest.mark.parametrize('name', ['jacobi_cfg2_subset', 'jacobi_cfg2_high_subset', 'jacobi_cfg3_subset'])
def test_jacobi_config_sized_vs_reference_golden(dev, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    F, omega, iters = int(g['F']), int(g['omega']), int(g['iters'])
    taps, lam, inv_on = _coeffs(F, int(g['width']), int(g['height']), int(g['definition']), g['hom'], omega)
    b = np.ascontiguousarray(g['inputs'].reshape(F, -1))
    x = _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters).reshape(g['outputs'].shape)
    assert np.abs(x - g['outputs']).max() <= 1e-9 * max(1.0, np.abs(g['outputs']).max())   # bar: 1e-4


@pytest.mark.parametrize('F,S,omega,iters', [
    (300, 578, 10, 100),      # cfg2, specialised <10,5>
    (320, 6, 10, 7), (321, 6, 10, 7), (640, 4, 10, 5), (1200, 4, 10, 20), (2400, 2, 10, 10),
    (600, 10, 30, 20), (1216, 2, 30, 5), (2432, 2, 30, 3),
    (1, 3, 10, 4), (2, 3, 10, 4), (11, 5, 10, 30),
    (50, 7, 5, 13), (3000, 2, 10, 3),
    # radii without a specialised kernel: the run-time-radius kernel (1 / 2 / 4 / 8 wavefronts per series, 3 / 5 / 7 / 19 frames per
    # lane, tap counts that are and are not multiples of the window length) ...
    (17, 3, 1, 9), (300, 578, 7, 20), (300, 100, 12, 20), (300, 600, 25, 10), (700, 3, 40, 6), (448, 600, 3, 8), (449, 600, 3, 8),
    (900, 40, 12, 6), (1800, 30, 7, 5), (3584, 3, 17, 3), (3585, 3, 17, 3), (9000, 2, 9, 2), (300, 20, 64, 5), (64, 5, 100, 4),
    (10000, 2, 7, 2),                                                    # ... and the last resort beyond its LDS
])
def test_jacobi_bit_exact_vs_c_oracle(dev, F, S, omega, iters):
    from meshflow_amd import synthetic
    from oracle import clib
    n = np.arange(F * S, dtype=np.int64).reshape(F, S)
    b = np.cumsum(3.0 * synthetic.normal(n, seed=F + S), axis=0)
    d = np.arange(-omega, omega + 1)
    taps = np.exp(-np.square((3 / omega) * d))
    lam = 0.2 + 0.75 * synthetic.uniform01(np.arange(F), seed=9)
    inv_on = 1.0 / (1 + 2 * lam * taps.sum())
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters)
    got = _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('omega', list(range(1, 34)) + [40, 47])
def test_jacobi_every_radius_bit_exact_vs_c_oracle(dev, omega):
    """mfs.py:46 accepts any temporal_smoothing_radius: radii 1..32 have kernels specialised ahead of time (csrc/jacobi_spec.hip),
    larger ones the run-time-radius kernel; short and long clips, one and several wavefronts per series -- all bit-identical."""
    from meshflow_amd import synthetic
    from oracle import clib
    for F, S, iters in ((300, 40, 12), (37, 9, 5), (1500, 3, 3)):
        b = np.cumsum(2.0 * synthetic.normal(np.arange(F * S).reshape(F, S), seed=omega + F), axis=0)
        taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
        lam = 0.1 + 0.85 * synthetic.uniform01(np.arange(F), seed=omega)
        inv_on = 1.0 / (1 + 2 * lam * taps.sum())
        want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
        got = _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters)
        assert np.array_equal(got, want), (omega, F, S)


@pytest.mark.parametrize('F,S,omega,iters,symmetric', [
    (600, 1030, 30, 5, True), (600, 1030, 30, 5, False),     # 1024 SIMDs + 6: the remainder is cut into 4 wavefronts per series on a side stream
    (300, 1100, 25, 4, True), (640, 2178, 30, 3, True),      # config 3's series count
    (600, 40, 30, 6, False), (700, 9, 23, 5, False), (300, 20, 22, 5, True),       # asymmetric taps: the full table (no symmetric fast path)
    (769, 1030, 30, 3, True),                                # too long for the 3-frames-per-lane remainder kernel: plain launch
])
def test_jacobi_wide_radius_tail_split_and_tap_symmetry(dev, F, S, omega, iters, symmetric):
    """Radii beyond 22 keep the OMEGA + 1 distinct values of SYMMETRIC taps in scalar registers (what mfs.py:750-752 always
    builds) and fall back to the full table for anything else; a series count just above a multiple of the chip's SIMDs sends the
    remainder to a second, split launch.  All bit-identical to the C oracle."""
    from meshflow_amd import synthetic
    from oracle import clib
    b = np.cumsum(2.0 * synthetic.normal(np.arange(F * S).reshape(F, S), seed=F + S), axis=0)
    taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
    if not symmetric:
        taps = taps * (1.0 + 0.3 * synthetic.uniform01(np.arange(2 * omega + 1), seed=5))
        assert not np.array_equal(taps, taps[::-1])
    lam = 0.1 + 0.85 * synthetic.uniform01(np.arange(F), seed=omega)
    inv_on = 1.0 / (1 + 2 * lam * taps.sum())
    want = clib.jacobi_banded(b, taps, lam, inv_on, omega, iters, openmp=True)
    got = _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters)
    assert np.array_equal(got, want)


def test_jacobi_full_cfg3_properties(dev):
    """Full config-3 size (F=600, 32x32 mesh, omega=30, 200 sweeps): linearity and constant-path fixed point."""
    from meshflow_amd import synthetic
    F, R, C, omega, iters = 600, 32, 32, 30, 200
    disp, hom = synthetic.motion(F, R, C, seed=0)
    taps, lam, inv_on = _coeffs(F, 1920, 1080, 0, hom, omega)
    b = disp.reshape(F, -1)
    x1 = _hip_jacobi(dev, b, taps, lam, inv_on, omega, iters)
    x2 = _hip_jacobi(dev, 2.0 * b, taps, lam, inv_on, omega, iters)
    np.testing.assert_array_equal(x2, 2.0 * x1)                 # scaling by 2 is exact in binary fp
    assert np.isfinite(x1).all()
    # smoothing reduces the high-frequency content of every path
    assert np.abs(np.diff(x1, n=2, axis=0)).mean() < 0.5 * np.abs(np.diff(b, n=2, axis=0)).mean()


def test_jacobi_host_wrapper(dev):
    from meshflow_amd import _lib, synthetic
    from oracle import clib
    F, S, omega, iters = 64, 10, 10, 12
    b = np.ascontiguousarray(synthetic.normal(np.arange(F * S).reshape(F, S), 4))
    taps = np.exp(-np.square((3 / omega) * np.arange(-omega, omega + 1)))
    lam = np.full(F, 0.5)
    inv_on = np.full(F, 0.1)
    x = np.empty_like(b)
    ms = ctypes.c_float(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(_lib.lib.mf_jacobi_f64_host(p(b), p(x), p(taps), p(lam), p(inv_on), F, S, omega, iters, ctypes.byref(ms)))
    np.testing.assert_array_equal(x, clib.jacobi_banded(b, taps, lam, inv_on, omega, iters))
    assert ms.value > 0


# ---------------------------------------------------------------------------------------------- warp

def _clip(F, H, W, R, C, seed, kind='noise', omega=3, iters=10, **kw):
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind=kind, **kw)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, omega, iters)
    return frames, disp, stab


def _hip_warp(dev, frames, R, C, unstab, stab, border=(0, 0, 255)):
    from meshflow_amd import ops
    n, H, W = frames.shape[:3]
    d_fr = torch.from_numpy(frames).to(dev)
    table = ops.cell_table(torch.from_numpy(np.ascontiguousarray(unstab)).to(dev),
                           torch.from_numpy(np.ascontiguousarray(stab)).to(dev), W, H, R, C)
    out = ops.warp(d_fr, table, border)
    torch.cuda.synchronize()
    table.check()
    return out.cpu().numpy(), table.crop.cpu().numpy(), table.records().cpu().numpy()


@pytest.mark.parametrize('H,W,R,C,kw', [
    (64, 96, 4, 4, dict(jitter_sigma=1.0)),
    (150, 200, 3, 5, dict(jitter_sigma=0.5)),                       # R != C, H not a multiple of 16
    (75, 101, 3, 5, dict(jitter_sigma=0.5)),                        # W % 4 != 0: byte-store path
    (130, 260, 16, 16, dict(jitter_sigma=0.3)),                     # small cells, 256 of them
    (96, 160, 32, 32, dict(translation_sigma=1.0, field_sigma=0.3)),  # 1024 cells
    (360, 640, 16, 16, dict(translation_sigma=8.0, jitter_sigma=2.0)),   # demo-video size, strong motion
    (272, 480, 3, 4, dict(jitter_sigma=0.5)),                       # cells of 120 x 90 pixels: mostly the plan-certified hot and pair paths
    (272, 480, 2, 2, dict(translation_sigma=3.0, jitter_sigma=1.5)),
])
def test_warp_bit_exact_vs_c_oracle(dev, H, W, R, C, kw):
    from oracle import clib
    frames, disp, stab = _clip(5, H, W, R, C, seed=H + W, **kw)
    out, crop, rec = _hip_warp(dev, frames, R, C, disp, stab)
    for f in range(frames.shape[0]):
        table, bad = clib.cell_table(W, H, R, C, disp[f], stab[f])
        assert bad == 0
        np.testing.assert_array_equal(rec[f], table)                      # homographies, rects, boxes
        want, want_crop = clib.warp_frame(frames[f], R, C, table)          # brute-force owner search
        np.testing.assert_array_equal(out[f], want)
        np.testing.assert_array_equal(crop[f], want_crop)


def test_warp_clip_cut_into_several_launches(dev, monkeypatch):
    """launch_warp cuts a clip whose table offsets would not fit 32 bits into several launches (frames, tables and crop rows
    offset per launch); MF_WARP_FRAMES_PER_LAUNCH forces the cut on a small clip: same bytes, same crop rows."""
    frames, disp, stab = _clip(7, 136, 256, 3, 4, seed=11, jitter_sigma=0.7)
    whole = _hip_warp(dev, frames, 3, 4, disp, stab)
    monkeypatch.setenv('MF_WARP_FRAMES_PER_LAUNCH', '3')
    cut = _hip_warp(dev, frames, 3, 4, disp, stab)
    np.testing.assert_array_equal(cut[0], whole[0])
    np.testing.assert_array_equal(cut[1], whole[1])
    from oracle import clib
    want, want_crop, bad = clib.warp_clip(frames, 3, 4, disp, stab)
    assert bad == 0
    np.testing.assert_array_equal(cut[0], want)
    np.testing.assert_array_equal(cut[1], want_crop)


@pytest.mark.parametrize('H,W,R,C,sigma,seed', [
    (130, 260, 4, 4, 8.0, 1),       # strongly non-affine quads: generic division path, wide reach
    (130, 260, 4, 4, 20.0, 2),      # folded quads: projective denominators change sign inside the frame
    (64, 96, 8, 8, 6.0, 3),         # more than 8 candidate cells per footprint: range-scan path
    (200, 300, 64, 64, 0.4, 4),     # largest supported mesh
    (17, 23, 2, 3, 1.0, 5),         # frame smaller than one tile
])
def test_warp_stress_geometries_bit_exact(dev, H, W, R, C, sigma, seed):
    """iid vertex jitter far beyond what a smoothed path produces: exercises the rare paths (irregular cells,
    overflowing candidate plans, generic division, border handling).  Degenerate cells are tolerated only
    if the oracle reports them too."""
    from meshflow_amd import synthetic
    from oracle import clib
    frames = synthetic.frames_numpy(2, H, W, seed=seed, kind='noise')
    n = np.arange(2 * (R + 1) * (C + 1) * 2, dtype=np.int64).reshape(2, R + 1, C + 1, 2)
    unstab = np.zeros((2, R + 1, C + 1, 2))
    stab = sigma * synthetic.normal(n, seed=100 + seed)
    want, want_crop, bad = clib.warp_clip(frames, R, C, unstab, stab)
    if bad:
        with pytest.raises(ValueError, match='degenerate'):
            _hip_warp(dev, frames, R, C, unstab, stab)
        return
    out, crop, rec = _hip_warp(dev, frames, R, C, unstab, stab)
    for f in range(2):
        table, _ = clib.cell_table(W, H, R, C, unstab[f], stab[f])
        np.testing.assert_array_equal(rec[f], table)
    assert np.array_equal(out, want), f'{(out != want).sum()} bytes differ'
    np.testing.assert_array_equal(crop, want_crop)


@pytest.mark.parametrize('H,W,R,C', [(24, 32764, 1, 64), (40, 16384, 2, 64), (16388, 32, 64, 1), (20, 8196, 1, 3),
                                     (2, 32767, 1, 1), (32767, 3, 64, 1), (9, 32767, 2, 33), (32767, 4, 9, 1)])       # (the limits themselves, thin)
def test_warp_extreme_aspect_ratios_bit_exact(dev, H, W, R, C):
    """Frames at the size limits (coordinates up to 32767: staging offsets, float32 edge margins that scale with the frame,
    15-bit region fields), very wide and very tall."""
    from oracle import clib
    frames, disp, stab = _clip(2, H, W, R, C, seed=H + W, kind='noise', jitter_sigma=0.6, translation_sigma=2.0)
    out, crop, rec = _hip_warp(dev, frames, R, C, disp, stab)
    want, want_crop, bad = clib.warp_clip(frames, R, C, disp, stab, use_bbox=True, openmp=True)
    assert bad == 0
    np.testing.assert_array_equal(crop, want_crop)
    assert np.array_equal(out, want), f'{(out != want).sum()} bytes differ'


def test_warp_matches_numpy_oracle_painter_loop(dev):
    """Against the reference-shaped per-cell painter loop (oracle/meshflow_oracle.py), small frame."""
    from oracle import meshflow_oracle as mo
    H, W, R, C = 48, 80, 4, 4
    frames, disp, stab = _clip(3, H, W, R, C, seed=11, jitter_sigma=1.5)
    out, crop, _ = _hip_warp(dev, frames, R, C, disp, stab, border=(10, 200, 30))
    want, bounds, per_frame = mo.stabilized_frames_and_crop_boundaries(list(frames), R, C, disp, stab, (10, 200, 30))
    np.testing.assert_array_equal(out, np.stack(want))
    np.testing.assert_array_equal(crop, per_frame)


def test_warp_known_answers(dev):
    from meshflow_amd import synthetic
    H, W, R, C = 64, 96, 4, 4
    frames = synthetic.frames_numpy(2, H, W, seed=1, kind='noise')
    z = np.zeros((2, R + 1, C + 1, 2))
    out, crop, _ = _hip_warp(dev, frames, R, C, z, z)
    np.testing.assert_array_equal(out, frames)                             # identity motion
    np.testing.assert_array_equal(crop, [[0, 0, W - 1, H - 1]] * 2)
    s = z.copy(); s[..., 0] = 5; s[..., 1] = -3                            # integer translation
    out, crop, _ = _hip_warp(dev, frames, R, C, z, s)
    want = np.empty_like(frames); want[...] = (0, 0, 255)
    want[:, 0:H - 3, 5:W] = frames[:, 3:H, 0:W - 5]
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(crop, [[5, 0, W - 1, H - 4]] * 2)
    s = z.copy(); s[..., 0] = 0.5                                          # half-pixel translation
    out, _, _ = _hip_warp(dev, frames, R, C, z, s)
    a = frames[:, :, 0:W - 1].astype(np.int64); b = frames[:, :, 1:W].astype(np.int64)
    np.testing.assert_array_equal(out[:, :, 1:W], ((16 * 32 * a + 16 * 32 * b + 512) >> 10).astype(np.uint8))


def test_warp_1080p_vs_c_oracle_and_identity(dev):
    """Config-2 geometry (1920x1080, 16x16 mesh): 3 frames against the C oracle, plus identity round trip."""
    from oracle import clib
    H, W, R, C = 1080, 1920, 16, 16
    frames, disp, stab = _clip(12, H, W, R, C, seed=0, kind='noise', omega=10, iters=100)
    sel = [3, 7, 11]
    out, crop, rec = _hip_warp(dev, frames[sel], R, C, disp[sel], stab[sel])
    want, want_crop, bad = clib.warp_clip(frames[sel], R, C, disp[sel], stab[sel], use_bbox=True, openmp=True)
    assert bad == 0
    np.testing.assert_array_equal(crop, want_crop)
    assert np.array_equal(out, want), f'{(out != want).sum()} bytes differ'
    z = np.zeros_like(disp[:2])
    out, crop, _ = _hip_warp(dev, frames[:2], R, C, z, z)
    assert np.array_equal(out, frames[:2])
    np.testing.assert_array_equal(crop, [[0, 0, W - 1, H - 1]] * 2)


@pytest.mark.parametrize('F,H,W,clip_frames,first', [(300, 1080, 1920, 300, 0), (150, 2160, 3840, 1200, 450)])
def test_warp_full_size_clip_properties(dev, F, H, W, clip_frames, first):
    """BASELINE config 2 at full size (300 frames of 1920x1080, 16x16 mesh) and config 4 at SHARD size (150 frames of
    3840x2160: frames 450..599 of the 1200-frame clip, what one of 8 GPUs holds; Jacobi over all 1200 frames), device-resident,
    through properties that do not need the oracle on every frame: an integer global shift moves every interior pixel by
    exactly that shift and paints the uncovered band in the border colour; the real smoothed motion is deterministic (two
    launches, same bytes) and a sample of frames is bit-identical to the C oracle; the clip-level bounds follow mfs.py:1103-1106."""
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib
    R, C = 16, 16
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern', first_frame=first)
    disp_all, hom = synthetic.motion(clip_frames, R, C, seed=0)
    d_disp_all = torch.from_numpy(disp_all).to(dev)
    disp, d_disp = disp_all[first:first + F], d_disp_all[first:first + F]
    # (a) integer shift: stabilized = unstabilized + (dx, dy) on every vertex of every frame
    dx, dy = 7, -5
    d_shift = d_disp + torch.tensor([dx, dy], dtype=torch.float64, device=dev)
    table = ops.cell_table(d_disp, d_shift, W, H, R, C)
    out = ops.warp(d_frames, table, (9, 8, 7))
    table.check()
    assert torch.equal(out[:, :H + dy, dx:], d_frames[:, -dy:, :W - dx])          # content moved by (+7, -5)
    border = torch.tensor([9, 8, 7], dtype=torch.uint8, device=dev)
    assert bool((out[:, :, :dx - 1] == border).all()) and bool((out[:, H + dy + 1:] == border).all())
    bounds = ops.crop_reduce(table.crop, W, H).tolist()
    assert bounds == [dx, 0, W - 1, H - 1 + dy]
    del out, table
    # (b) the smoothed motion of the config
    s = MeshFlowStabilizer(device=str(dev))
    d_stab = s._stabilized_vertex_displacements_device(d_disp_all, W, H, 0, hom)[first:first + F]
    out1, crop1 = s._stabilized_frames_device(d_frames, d_disp, d_stab)
    out1, crop1 = out1.clone(), crop1.clone()
    out2, crop2 = s._stabilized_frames_device(d_frames, d_disp, d_stab)
    assert torch.equal(out1, out2) and torch.equal(crop1, crop2)
    sel = [0, F // 2 - 1, F - 1]
    stab = d_stab.cpu().numpy()
    want, want_crop, bad = clib.warp_clip(d_frames[sel].cpu().numpy(), R, C, disp[sel], stab[sel], use_bbox=True, openmp=True)
    assert bad == 0
    assert np.array_equal(out1[sel].cpu().numpy(), want)
    np.testing.assert_array_equal(crop1[sel].cpu().numpy(), want_crop)
    crop_h = crop1.cpu().numpy()
    assert ops.crop_reduce(crop1, W, H).tolist() == [crop_h[:, 0].max(), crop_h[:, 1].max(), crop_h[:, 2].min(), crop_h[:, 3].min()]


def test_warp_4k_frames_vs_c_oracle(dev):
    """BASELINE config 4 geometry (3840x2160, 16x16 mesh): two frames of a shard, bit-identical to the C oracle."""
    from oracle import clib
    H, W, R, C = 2160, 3840, 16, 16
    frames, disp, stab = _clip(6, H, W, R, C, seed=4, kind='pattern', omega=10, iters=50)
    sel = [2, 5]
    out, crop, _ = _hip_warp(dev, frames[sel], R, C, disp[sel], stab[sel])
    want, want_crop, bad = clib.warp_clip(frames[sel], R, C, disp[sel], stab[sel], use_bbox=True, openmp=True)
    assert bad == 0
    np.testing.assert_array_equal(crop, want_crop)
    assert np.array_equal(out, want), f'{(out != want).sum()} bytes differ'


def test_trimmed_reciprocal_matches_ieee_division(dev):
    from meshflow_amd import _lib
    bad = ctypes.c_uint64(123)
    _lib.check(_lib.lib.mf_selftest_recip(1 << 32, 12345, ctypes.byref(bad)))
    assert bad.value == 0


def test_fast_coordinate_chain_never_differs_unflagged(dev):
    """The hot path's cheap float64 chain (fused affine forms, reciprocal to an ulp) must give the float32 coordinates of
    cv2.perspectiveTransform's own arithmetic whenever its midpoint guard stays silent: 2^31 hashed (matrix, position) cases that
    satisfy the plan's premises, a quarter of them at the certified limits."""
    from meshflow_amd import _lib
    c = (ctypes.c_uint64 * 3)(1, 1, 1)
    _lib.check(_lib.lib.mf_selftest_fast64(1 << 31, 777, c))
    missed, flagged, tested = c[0], c[1], c[2]
    assert tested > (1 << 31)                  # most draws satisfy the premises (8 values per case)
    assert missed == 0
    assert flagged / tested < 2e-5             # guard window +-512 of 2^29 low-mantissa patterns: ~2e-6 per value
    # ... and the cheap values themselves stay far inside the window: the certified bound is 118 float64 ulps
    far = ctypes.c_double(-1.0)
    _lib.check(_lib.lib.mf_selftest_fast64_margin(1 << 28, 4242, ctypes.byref(far)))
    assert 0.0 < far.value <= 118.0, far.value


def test_degenerate_mesh_is_reported(dev):
    from meshflow_amd import synthetic
    H, W, R, C = 64, 96, 4, 4
    frames = synthetic.frames_numpy(1, H, W, seed=1)
    z = np.zeros((1, R + 1, C + 1, 2))
    s = z.copy()
    grid_x = np.array([np.ceil((W - 1) * c / C) for c in range(C + 1)])
    s[0, :, :, 0] = -grid_x[None, :]                   # collapse every vertex onto x = 0
    with pytest.raises(ValueError, match='degenerate'):
        _hip_warp(dev, frames, R, C, z, s)


@pytest.mark.parametrize('moved,onto', [('TL', ('TR', 'BL')), ('TL', ('TR', 'BR')), ('BR', ('TR', 'BL')), ('TR', ('TL', 'BR'))])
def test_three_collinear_corners_are_degenerate_whichever_corner(dev, moved, onto):
    """one cell of one frame gets a corner moved onto the line through two others (the triple may include the TL corner: the
    closed-form solver's own denominator only sees TR, BR, BL): the cell table counts it, the C oracle flags the same cells."""
    from meshflow_amd import ops
    from oracle import clib
    H, W, R, C = 96, 128, 4, 4
    gx = np.array([np.ceil((W - 1) * c / C) for c in range(C + 1)])
    gy = np.array([np.ceil((H - 1) * r / R) for r in range(R + 1)])
    pos = {'TL': (1, 1), 'TR': (1, 2), 'BL': (2, 1), 'BR': (2, 2)}                      # (row, col) of the corners of cell (1, 1)
    z = np.zeros((1, R + 1, C + 1, 2))
    s = z.copy()
    (ra, ca), (rb, cb), (rm, cm) = pos[onto[0]], pos[onto[1]], pos[moved]
    a, b = np.array([gx[ca], gy[ra]]), np.array([gx[cb], gy[rb]])
    target = a + 0.5 * (b - a)
    s[0, rm, cm] = target - np.array([gx[cm], gy[rm]])                                  # stabilized = grid + (stab - unstab)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    table = ops.cell_table(t(z), t(s), W, H, R, C)
    torch.cuda.synchronize()
    bad = int(table.status.item())
    _, want_bad = clib.cell_table(W, H, R, C, z[0], s[0])
    assert bad == want_bad and bad >= 1
    with pytest.raises(ValueError, match='degenerate'):
        table.check()


def test_stabilizer_class_end_to_end(dev):
    """The drop-in boundary with the reference's own signatures (host buffers in and out)."""
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib, meshflow_oracle as mo
    F, H, W, R, C = 24, 96, 128, 4, 4
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=2, kind='pattern', jitter_sigma=0.5)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=5,
                           optimization_num_iterations=20)
    out_frames, bounds, stab, score = s.stabilize_clip(list(frames), disp, hom)
    want_stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 5, 20)
    assert np.abs(stab - want_stab).max() < 1e-9
    assert score == mo.stability_score(stab)
    want, want_crop, _ = clib.warp_clip(frames, R, C, disp, stab)
    assert isinstance(out_frames, list) and len(out_frames) == F and out_frames[0].shape == (H, W, 3)
    np.testing.assert_array_equal(np.stack(out_frames), want)
    assert tuple(int(v) for v in bounds) == (want_crop[:, 0].max(), want_crop[:, 1].max(),
                                             want_crop[:, 2].min(), want_crop[:, 3].min())
    # the two private methods with the reference's own signatures give the same arrays
    np.testing.assert_array_equal(s._get_stabilized_vertex_displacements(F, list(frames), 0, disp, hom), stab)
    fr2, b2 = s._get_stabilized_frames_and_crop_boundaries(F, list(frames), disp, stab)
    np.testing.assert_array_equal(np.stack(fr2), want)
    assert tuple(int(v) for v in b2) == tuple(int(v) for v in bounds)
    # crop=True: device-resident crop + resize of the stabilized frames
    res = s.stabilize_clip(list(frames), disp, hom, crop=True, keep_uncropped=False)
    assert res[0] is None and len(res) == 5
    np.testing.assert_array_equal(np.stack(res[4]), np.stack(mo.crop_frames(list(want), bounds)))


@pytest.mark.parametrize('chunk_frames,io_threads,slots', [(1, 1, 2), (1, 4, 3), (7, 2, 2), (5, 4, 8), (2, 3, 4), (64, 3, 8)])
def test_chunked_staging_gives_the_same_clip(dev, monkeypatch, chunk_frames, io_threads, slots):
    """The host pipeline (csrc/hostpipe.hip): any chunking, thread count and ring depth (MF_PIPE_CHUNK / _UP / _DOWN / _SLOTS, read at
    every call), list or array input, every output identical -- ragged last chunk, more threads than chunks, one frame per chunk, a
    ring of two slots that every chunk after the second has to wait for."""
    from meshflow_amd import synthetic
    monkeypatch.setenv('MF_PIPE_CHUNK', str(chunk_frames))
    monkeypatch.setenv('MF_PIPE_UP', str(io_threads))
    monkeypatch.setenv('MF_PIPE_DOWN', str(io_threads))
    monkeypatch.setenv('MF_PIPE_SLOTS', str(slots))
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib, meshflow_oracle as mo
    F, H, W, R, C = 23, 72, 100, 3, 5
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=9, kind='noise', jitter_sigma=0.7)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=4, optimization_num_iterations=15)
    separate = [f.copy() for f in frames]
    for inp in (separate, frames):
        got = s.stabilize_clip(inp, disp, hom, crop=True)
        out, bounds, stab, score, cropped = got
        want, want_crop, _ = clib.warp_clip(frames, R, C, disp, stab)
        np.testing.assert_array_equal(np.stack(out), want)
        assert tuple(int(v) for v in bounds) == (want_crop[:, 0].max(), want_crop[:, 1].max(),
                                                 want_crop[:, 2].min(), want_crop[:, 3].min())
        np.testing.assert_array_equal(np.stack(cropped), np.stack(mo.crop_frames(list(want), bounds)))
        # the two methods of the reference's call sequence by themselves (mfs.py:154, 159), through the same ring
        out2, bounds2 = s._get_stabilized_frames_and_crop_boundaries(F, inp, disp, stab)
        np.testing.assert_array_equal(np.stack(out2), want)
        assert tuple(int(v) for v in bounds2) == tuple(int(v) for v in bounds)
        np.testing.assert_array_equal(np.stack(s._crop_frames(out2, bounds2)), np.stack(cropped))


def test_chunked_staging_reports_a_bad_frame(dev):
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F, H, W, R, C = 9, 48, 64, 2, 2
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=1)
    bad = [f.copy() for f in frames]
    bad[6] = np.zeros((H, W + 1, 3), np.uint8)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=2, optimization_num_iterations=5)
    with pytest.raises(ValueError, match='shape'):
        s.stabilize_clip(bad, disp, hom)
    out, *_ = s.stabilize_clip([f.copy() for f in frames], disp, hom)    # still usable
    assert len(out) == F


def test_warp_host_wrapper(dev):
    from meshflow_amd import _lib
    from oracle import clib
    H, W, R, C = 64, 96, 4, 4
    frames, disp, stab = _clip(3, H, W, R, C, seed=5, jitter_sigma=1.0)
    out = np.empty_like(frames)
    crop = np.zeros((3, 4), np.int32)
    border = (ctypes.c_uint8 * 3)(0, 0, 255)
    ms = ctypes.c_float(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(_lib.lib.mf_warp_u8c3_host(p(frames), p(out), p(np.ascontiguousarray(disp)), p(np.ascontiguousarray(stab)),
                                          3, W, H, R, C, border, p(crop), ctypes.byref(ms)))
    want, want_crop, _ = clib.warp_clip(frames, R, C, disp, stab)
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(crop, want_crop)


# ---------------------------------------------------------------------------------------------- crop + resize

@pytest.mark.parametrize('H,W,bounds', [
    (48, 64, (5, 3, 60, 44)), (48, 64, (0, 0, 63, 47)), (75, 101, (7, 9, 90, 70)), (40, 60, (10, 10, 10, 10)),
    (360, 640, (13, 11, 629, 350)), (33, 31, (1, 2, 29, 30)),
])
def test_crop_resize_bit_exact_vs_oracle(dev, H, W, bounds):
    from meshflow_amd import ops, synthetic
    from oracle import meshflow_oracle as mo
    frames = synthetic.frames_numpy(3, H, W, seed=H, kind='noise')
    got = ops.crop_resize(torch.from_numpy(frames).to(dev), bounds).cpu().numpy()
    want = np.stack(mo.crop_frames(list(frames), bounds))
    np.testing.assert_array_equal(got, want)


def test_crop_resize_1080p_and_errors(dev):
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import meshflow_oracle as mo
    frames = synthetic.frames_numpy(2, 1080, 1920, seed=5, kind='noise')
    bounds = (13, 11, 1909, 1068)
    got = MeshFlowStabilizer()._crop_frames(list(frames), bounds)
    want = mo.crop_frames(list(frames), bounds)
    assert isinstance(got, list) and len(got) == 2
    np.testing.assert_array_equal(np.stack(got), np.stack(want))
    with pytest.raises(ValueError):
        ops.crop_resize(torch.from_numpy(frames[:1]).to(dev), (50, 10, 40, 100))       # right < left: empty crop


def test_crop_resize_full_cfg2_clip_properties(dev):
    """300 frames of 1920x1080 on the device: the full-frame rectangle is the identity (weights (2048, 0) both ways), a
    real crop is deterministic and a sample of frames matches the oracle."""
    from meshflow_amd import ops, synthetic
    from oracle import meshflow_oracle as mo
    F, H, W = 300, 1080, 1920
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=3, kind='noise')
    assert torch.equal(ops.crop_resize(d_frames, (0, 0, W - 1, H - 1)), d_frames)
    rect = (13, 11, 1909, 1068)
    out = ops.crop_resize(d_frames, rect)
    assert torch.equal(out, ops.crop_resize(d_frames, rect))
    sel = [0, 150, 299]
    want = np.stack(mo.crop_frames(list(d_frames[sel].cpu().numpy()), rect))
    assert np.array_equal(out[sel].cpu().numpy(), want)


def test_randomised_parity_campaign(dev):
    """300 random geometries (warp incl. W % 4 != 0, borders, folded quads; Jacobi incl. the generic kernel;
    crop + resize) against the C / NumPy oracles, bit for bit.  Seed 1 contains the case that exposed the
    crop-flag bug for pixels beyond the right frame edge."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fuzz_parity.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, stats = mod.run(300, 1)
    assert bad == 0 and stats['warp'] > 100 and stats['jacobi'] > 30 and stats['resize'] > 30


def test_methods_borrowed_by_a_foreign_class(dev):
    """a class that only has the reference's attributes borrows the drop-in methods."""
    from meshflow_amd import synthetic
    import meshflow_amd as amd
    from oracle import clib

    class RefLike:                       # stands in for meshflowstabilizer.MeshFlowStabilizer (attributes of mfs.py:87-97)
        def __init__(self):
            self.mesh_row_count = self.mesh_col_count = 4
            self.temporal_smoothing_radius, self.optimization_num_iterations = 3, 8
            self.color_outside_image_area_bgr = (0, 0, 255)

    class Stabilizer(RefLike):
        _get_stabilized_vertex_displacements = amd.MeshFlowStabilizer._get_stabilized_vertex_displacements
        _get_stabilized_frames_and_crop_boundaries = amd.MeshFlowStabilizer._get_stabilized_frames_and_crop_boundaries
        _check_mesh_shape = amd.MeshFlowStabilizer._check_mesh_shape
        _torch_device = amd.MeshFlowStabilizer._torch_device
        _jacobi_coefficients_device = amd.MeshFlowStabilizer._jacobi_coefficients_device
        _stabilized_vertex_displacements_device = amd.MeshFlowStabilizer._stabilized_vertex_displacements_device
        _crop_frames = amd.MeshFlowStabilizer._crop_frames
        device = None

    F, H, W = 10, 64, 96
    frames, disp, hom = synthetic.clip(F, H, W, 4, 4, seed=6)
    s = Stabilizer()
    stab = s._get_stabilized_vertex_displacements(F, list(frames), 0, disp, hom)
    out, bounds = s._get_stabilized_frames_and_crop_boundaries(F, list(frames), disp, stab)
    want, crop, _ = clib.warp_clip(frames, 4, 4, disp, stab)
    np.testing.assert_array_equal(np.stack(out), want)
    cropped = s._crop_frames(out, bounds)
    assert len(cropped) == F and cropped[0].shape == (H, W, 3)


@pytest.mark.parametrize('F,R,C', [(300, 16, 16), (48, 4, 4), (7, 2, 3), (601, 32, 32), (6, 2, 2), (4, 3, 2), (3, 2, 2), (2, 1, 1)])
def test_stability_score_on_device_vs_host(dev, F, R, C):
    """mf_stability_score_f64 (five direct DFT bins + Parseval) against the reference's np.fft formulation (host.py,
    pinned against the reference's own value in tests/test_oracle_golden.py): float64 rounding apart."""
    from meshflow_amd import host, ops, synthetic
    disp, _ = synthetic.motion(F, R, C, seed=F, jitter_sigma=0.3)
    score, series = ops.stability_score(torch.from_numpy(disp).to(dev))
    want = host.stability_score(disp)
    assert abs(float(score.item()) - want) <= 1e-12
    xs = np.diff(disp.reshape(F, -1), axis=0).T                                # (S, N)
    en = np.square(np.abs(np.fft.fft(xs)))
    np.testing.assert_allclose(series.cpu().numpy(), en[:, 1:6].sum(1) / en.sum(1), rtol=1e-10, atol=1e-14)
    with pytest.raises(ValueError):                                           # F = 1: np.fft.fft of an empty profile raises too
        ops.stability_score(torch.zeros((1, 3, 3, 2), dtype=torch.float64, device=dev))


def test_tiny_frames_equal_the_oracle(dev):
    """Frames of 2-100 columns and 2-40 rows (a 4-column frame has no "deep interior" pixel at all: the bound of that test used to wrap
    around as an unsigned number and sent its footprints down the unclamped path -- found by this sweep at the end of round 5), meshes
    of 1-3 rows and columns, still and moving: frames, per-frame crop values and the degenerate-cell count as the C oracle's; the
    scan-only kernel agrees with the fused scan."""
    import itertools
    import torch
    from meshflow_amd import ops
    from oracle import clib
    rng = np.random.default_rng(11)
    n = 0
    for W, H, (R, C), nfr in itertools.product((2, 3, 4, 5, 6, 7, 8, 12, 31, 32, 33, 64, 100), (2, 3, 4, 5, 8, 9, 12, 13, 17, 40),
                                               ((1, 1), (1, 2), (2, 1), (2, 3), (3, 3)), (1, 3)):
        if C > W - 1 or R > H - 1:
            continue
        frames = rng.integers(0, 256, size=(nfr, H, W, 3), dtype=np.uint8)
        unstab = np.zeros((nfr, R + 1, C + 1, 2))
        scale = rng.choice([0.0, 0.3, 1.5, 4.0])
        stab = rng.normal(0, 1, size=(nfr, 1, 1, 2)) * scale + rng.normal(0, 0.2, size=(nfr, R + 1, C + 1, 2)) * min(scale, 1.0)
        want, want_crop, want_bad = clib.warp_clip(frames, R, C, unstab, stab, (9, 8, 7))
        table = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
        out = ops.warp(torch.from_numpy(frames).to(dev), table, (9, 8, 7))
        assert int(table.status.item()) == want_bad, (W, H, R, C, nfr)
        if want_bad:
            continue
        n += 1
        assert np.array_equal(out.cpu().numpy(), want), (W, H, R, C, nfr, float(scale))
        assert np.array_equal(table.crop.cpu().numpy(), want_crop), (W, H, R, C, nfr, float(scale))
        table2 = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
        ops.crop_scan(table2)
        assert np.array_equal(table2.crop.cpu().numpy(), want_crop), ('scan', W, H, R, C, nfr, float(scale))
    assert n > 800
    # meshes finer than the pixel grid: vertex coordinates repeat (mfs.py:881-906 rounds them up), the cells between them have no
    # homography -- the reference would die inside cv2; here the count of such cells is the oracle's and nothing faults
    for (W, H), (R, C) in itertools.product(((2, 2), (3, 5), (4, 4), (8, 3), (16, 16), (33, 20)), ((2, 2), (4, 4), (8, 16), (16, 8), (40, 40), (64, 64))):
        frames = rng.integers(0, 256, size=(2, H, W, 3), dtype=np.uint8)
        unstab = np.zeros((2, R + 1, C + 1, 2))
        stab = rng.normal(0, 0.3, size=(2, R + 1, C + 1, 2))
        want, want_crop, want_bad = clib.warp_clip(frames, R, C, unstab, stab, (1, 2, 3))
        table = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
        out = ops.warp(torch.from_numpy(frames).to(dev), table, (1, 2, 3))
        assert int(table.status.item()) == want_bad, (W, H, R, C)
        if not want_bad:
            assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(table.crop.cpu().numpy(), want_crop), (W, H, R, C)


def test_crop_resize_and_score_corner_cases(dev):
    """`_crop_frames` (mfs.py:1111-1157) on frames of 1-300 columns and 1-40 rows with extreme rectangles (one pixel, one row, one column,
    the whole frame, random ones): 2,000+ cases equal to the NumPy oracle.  (A stack of ONE pixel used to read the byte in front of its
    allocation: found by this sweep at the end of round 5.)  The stability score (mfs.py:1216-1259) on clips of 1-65 frames."""
    import itertools
    import torch
    from meshflow_amd import ops
    from oracle import meshflow_oracle as mo
    rng = np.random.default_rng(5)
    n = 0
    for W, H, nfr in itertools.product((1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 100, 255, 256, 257, 300), (1, 2, 3, 5, 8, 9, 17, 33, 40), (1, 3)):
        frames = rng.integers(0, 256, size=(nfr, H, W, 3), dtype=np.uint8)
        rects = {(0, 0, W - 1, H - 1), (0, 0, 0, 0), (W - 1, H - 1, W - 1, H - 1), (0, H - 1, W - 1, H - 1), (W - 1, 0, W - 1, H - 1)}
        for _ in range(3):
            l, r = sorted(rng.integers(0, W, size=2))
            t, b = sorted(rng.integers(0, H, size=2))
            rects.add((int(l), int(t), int(r), int(b)))
        d_frames = torch.from_numpy(frames).to(dev)
        # (and from / into stacks that do not start on a 4-byte boundary)
        raw_in = torch.zeros(frames.size + 8, dtype=torch.uint8, device=dev)
        raw_out = torch.zeros(frames.size + 8, dtype=torch.uint8, device=dev)
        src = raw_in[1:1 + frames.size].view(frames.shape)
        src.copy_(d_frames)
        dst = raw_out[3:3 + frames.size].view(frames.shape)
        for rect in sorted(rects):
            n += 1
            want = np.stack(mo.crop_frames(list(frames), rect))
            assert np.array_equal(ops.crop_resize(d_frames, rect).cpu().numpy(), want), (W, H, nfr, rect)
            assert np.array_equal(ops.crop_resize(src, rect, out=dst).cpu().numpy(), want), ('unaligned', W, H, nfr, rect)
    assert n > 1500
    for F, S in itertools.product((1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 33, 64, 65), (2, 8, 50)):
        stab = np.cumsum(rng.normal(size=(F, S // 2, 1, 2)), axis=0)
        d_stab = torch.from_numpy(np.ascontiguousarray(stab)).to(dev)
        try:
            want = mo.stability_score(stab)
        except ValueError:                                  # a clip of one frame has no velocity profile: np.fft refuses (mfs.py:1241)
            with pytest.raises(Exception):
                ops.stability_score(d_stab)
            continue
        score, _ = ops.stability_score(d_stab)
        got = float(score.item())
        assert (np.isnan(got) and np.isnan(want)) or abs(got - want) <= 1e-12 * max(1.0, abs(want)), (F, S, got, want)


def test_device_clip_beyond_four_gigabytes(dev):
    """720 frames of 1080p = 4.48 GB per stack: the byte offsets of frames 690+ lie beyond 2^32 inside ONE launch of the warp, the scan
    and the crop + resize.  The big launch equals small launches on the same frames (first, around the 4 GiB line, last), and the C
    oracle on the three frames around the line."""
    import torch
    from meshflow_amd import ops, synthetic
    from oracle import clib
    F, H, W, R, C = 720, 1080, 1920, 16, 16
    disp, _ = synthetic.motion(F, R, C, seed=3)
    stab = np.ascontiguousarray(disp + 0.8 * synthetic.normal(np.arange(disp.size).reshape(disp.shape), seed=9))
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=3)
    d_un, d_st = torch.from_numpy(disp).to(dev), torch.from_numpy(stab).to(dev)
    table = ops.cell_table(d_un, d_st, W, H, R, C)
    out = ops.warp(d_frames, table, (0, 0, 255))
    crop = table.crop.clone()
    table.check()
    for lo, hi in ((0, 3), (688, 693), (717, 720)):
        t2 = ops.cell_table(d_un[lo:hi], d_st[lo:hi], W, H, R, C)
        o2 = ops.warp(d_frames[lo:hi].contiguous(), t2, (0, 0, 255))
        assert torch.equal(o2, out[lo:hi]) and torch.equal(t2.crop, crop[lo:hi]), (lo, hi)
    want, want_crop, bad = clib.warp_clip(d_frames[689:692].cpu().numpy(), R, C, disp[689:692], stab[689:692], (0, 0, 255), use_bbox=True, openmp=True)
    assert bad == 0 and np.array_equal(out[689:692].cpu().numpy(), want) and np.array_equal(crop[689:692].cpu().numpy(), want_crop)
    rect = (13, 11, 1909, 1068)
    resized = ops.crop_resize(out, rect)
    for lo, hi in ((0, 2), (689, 692), (718, 720)):
        assert torch.equal(ops.crop_resize(out[lo:hi].contiguous(), rect), resized[lo:hi]), (lo, hi)
    scan_table = ops.cell_table(d_un, d_st, W, H, R, C)
    ops.crop_scan(scan_table)
    assert torch.equal(scan_table.crop, crop)


def test_non_finite_and_huge_paths_never_fault(dev):
    """NaN, +-inf, 1e300, 1e18, 1e7, -1e5 and a denormal in the stabilized paths -- at one vertex, in one frame, everywhere: the cell
    table counts exactly the cells the oracle counts as having no homography, the frames of the remaining ones equal the oracle's,
    and nothing faults (the reference would die inside cv2 on most of these)."""
    import torch
    from meshflow_amd import ops
    from oracle import clib
    rng = np.random.default_rng(2)
    H, W, R, C, n = 48, 64, 3, 4, 3
    frames = rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8)
    unstab = np.zeros((n, R + 1, C + 1, 2))
    for poison in (np.nan, np.inf, -np.inf, 1e300, 1e18, 1e7, -1e5, 5e-324):
        for where in range(3):
            stab = rng.normal(0, 0.5, size=(n, R + 1, C + 1, 2))
            if where == 0:
                stab[1, 2, 2, 0] = poison
            elif where == 1:
                stab[1] = poison
            else:
                stab[:] = poison
            want, want_crop, want_bad = clib.warp_clip(frames, R, C, unstab, stab, (1, 2, 3))
            table = ops.cell_table(torch.from_numpy(unstab).to(dev), torch.from_numpy(stab).to(dev), W, H, R, C)
            out = ops.warp(torch.from_numpy(frames).to(dev), table, (1, 2, 3))
            torch.cuda.synchronize()
            assert int(table.status.item()) == want_bad, (poison, where)
            if not want_bad:
                assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(table.crop.cpu().numpy(), want_crop), (poison, where)


def test_unaligned_frame_stacks_repeatedly(dev):
    """Frame stacks that do not start on a 4-byte boundary (odd frame sizes cut into frame ranges, a caller's slice) run
    `warp_kernel<false>`.  Until the end of round 5 that instantiation still issued the hot path's speculative matrix load: its
    destination registers were dead there, the compiler reused them for the lane mask while the load was in flight, and about one
    launch in 200 left rows unwritten.  600 launches from an odd address and every chunking of `mf_warp_clip_u8c3` on 9 x 17 pixel
    frames: every byte as the aligned launch wrote it."""
    import itertools
    import torch
    from meshflow_amd import ops, synthetic
    n, H, W, R, C = 33, 17, 9, 1, 1
    frames, disp, _ = synthetic.clip(n, H, W, R, C, seed=50, kind='noise', jitter_sigma=0.8)
    stab = np.ascontiguousarray(disp + 0.5 * synthetic.normal(np.arange(disp.size).reshape(disp.shape), seed=33))
    d_un, d_st = torch.from_numpy(disp).to(dev), torch.from_numpy(stab).to(dev)
    d_fr = torch.from_numpy(frames).to(dev)
    table = ops.cell_table(d_un, d_st, W, H, R, C)
    ref = ops.warp(d_fr, table, (4, 5, 6)).clone()
    table.check()
    raw_in = torch.zeros(d_fr.numel() + 8, dtype=torch.uint8, device=dev)
    raw_out = torch.zeros(d_fr.numel() + 8, dtype=torch.uint8, device=dev)
    for shift in (1, 2, 3):
        src = raw_in[shift:shift + d_fr.numel()].view(d_fr.shape)
        src.copy_(d_fr)
        dst = raw_out[shift:shift + d_fr.numel()].view(d_fr.shape)
        for _ in range(200):
            dst.fill_(0xEE)
            ops.warp(src, table, (4, 5, 6), out=dst)
            assert torch.equal(dst, ref), shift
    for trial, (chunks, own) in enumerate(itertools.product((1, 2, 3, 7, 33, 40), (False, True))):
        prep = torch.cuda.Stream(device=dev) if own else None
        for _ in range(60):
            t2 = ops.CellTable(n, W, H, R, C, dev)
            out = torch.full_like(d_fr, 0xEE)
            if prep is not None:
                prep.wait_stream(torch.cuda.current_stream())
            got, bounds = ops.warp_clip(d_fr, d_un, d_st, t2, (4, 5, 6), out=out, chunks=chunks, prep_stream=prep)
            torch.cuda.synchronize()
            assert torch.equal(got, ref) and torch.equal(t2.crop, table.crop), (chunks, own)

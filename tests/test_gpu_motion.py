"""Parity of the vertex-motion kernels (mfs.py:236-452 after the tracker) against the
reference's own outputs (tests/golden/motion_*.npz) and the C oracle, through the C ABI.  Needs an MI355X.
Everything on this row is selection or single IEEE operations, so the bar is bit-exact."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

CASES = ('motion_small', 'motion_1080p', 'motion_ragged')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _load_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    off = g['offsets']
    feats = [(g['early'][off[t]:off[t + 1]].reshape(-1, 1, 2), g['late'][off[t]:off[t + 1]].reshape(-1, 1, 2))
             for t in range(len(off) - 1)]
    return g, feats


def _stabilizer(g):
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    return MeshFlowStabilizer(mesh_row_count=int(g['R']), mesh_col_count=int(g['C']),
                              feature_ellipse_row_count=int(g['ell_rows']), feature_ellipse_col_count=int(g['ell_cols']))


def test_sqrt_is_correctly_rounded(dev):
    from meshflow_amd import _lib
    bad = ctypes.c_uint64(1)
    _lib.check(_lib.lib.mf_selftest_sqrt(1 << 28, 2024, ctypes.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize('name', CASES)
def test_displacements_vs_reference_golden(dev, golden_dir, name):
    g, feats = _load_case(golden_dir, name)
    s = _stabilizer(g)
    disp, hom = s._get_unstabilized_vertex_displacements_from_features(int(g['F']), int(g['width']), int(g['height']),
                                                                       feats, g['hom'])
    assert disp.dtype == np.float64 and disp.shape == g['displacements'].shape
    assert np.array_equal(disp, g['displacements'])
    assert np.array_equal(hom, g['hom'])


@pytest.mark.parametrize('name', CASES)
def test_velocities_vs_reference_golden(dev, golden_dir, name):
    g, feats = _load_case(golden_dir, name)
    s = _stabilizer(g)
    for t, (e, l) in enumerate(feats):
        vel = s._get_unstabilized_vertex_velocities_from_features(int(g['width']), int(g['height']), e, l, g['hom'][t])
        assert vel.dtype == np.float32 and np.array_equal(vel, g['velocities'][t])


@pytest.mark.parametrize('geometry', [
    (1920, 1080, 16, 16, 10, 10, 24, (700, 1100), 21),
    (1920, 1080, 32, 32, 10, 10, 6, (1500, 2500), 22),
    (3840, 2160, 16, 16, 10, 10, 5, (4200, 5200), 23),      # > 4096 features per pair: the sort leaves one LDS tile
    (3840, 2160, 16, 16, 10, 10, 3, (9000, 10000), 28),     # 16384 slots: two levels of global merge steps
    (321, 243, 5, 9, 7, 3, 9, (40, 90), 24),
    (640, 480, 12, 7, 2, 9, 7, (200, 260), 25),
    (200, 100, 1, 1, 10, 10, 4, (5, 9), 26),
    (640, 360, 8, 8, 40, 40, 4, (300, 500), 27),            # every ellipse covers the whole mesh
])
def test_vs_c_oracle(dev, geometry):
    from oracle import clib, gen_golden
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    W, H, R, C, er, ec, F, per_pair, seed = geometry
    feats, hom = gen_golden.motion_inputs(W, H, R, C, F, per_pair, seed)
    feats[1] = (None, None)                                   # a pair the tracker gave up on
    want_d, want_v = clib.vertex_motion(W, H, R, C, er, ec, feats, hom, openmp=True)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, feature_ellipse_row_count=er, feature_ellipse_col_count=ec)
    got_d, got_v = s._vertex_motion_from_features(F, W, H, feats, hom)
    assert np.array_equal(got_v, want_v)
    assert np.array_equal(got_d, want_d)


def test_full_cfg2_sized_clip_vs_c_oracle(dev):
    """299 frame pairs of a 1080p / 16x16 clip, 1500-2500 features each (590 k in all): bit-identical to the C oracle, and
    the running sum is the sum (property: displacement[t] - displacement[t-1] == float64(velocity[t-1]) exactly)."""
    from oracle import clib
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    W, H, R, C, F = 1920, 1080, 16, 16, 300
    _, hom = synthetic.motion(F, R, C, seed=0)
    hom[:-1, :2, :2] = np.identity(2) + 0.2 * (hom[:-1, :2, :2] - np.identity(2))
    feats = synthetic.features(F, H, W, hom, seed=0, per_pair=(1500, 2500))
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C)
    disp, vel = s._vertex_motion_from_features(F, W, H, feats, hom)
    want_d, want_v = clib.vertex_motion(W, H, R, C, 10, 10, feats, hom, openmp=True)
    assert np.array_equal(vel, want_v) and np.array_equal(disp, want_d)
    assert np.array_equal(np.diff(disp, axis=0), vel.astype(np.float64)) or np.abs(np.diff(disp, axis=0) - vel).max() < 1e-9
    assert not disp[0].any()


def test_duplicate_values_and_even_counts(dev):
    """Ties: many features with identical residuals, even list lengths (mean of the two middle values)."""
    from oracle import clib
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    W, H, R, C = 320, 240, 4, 4
    k = 64
    ex = np.tile(np.array([40.0, 160.0, 280.0, 100.0]), k // 4)
    ey = np.repeat(np.array([30.0, 120.0, 210.0, 60.0]), k // 4)
    early = np.stack([ex, ey], -1)
    late = early + np.array([0.5, -0.25]) * (np.arange(k) % 3)[:, None]       # three distinct residuals, many repeats
    feats = [(early.reshape(-1, 1, 2), late.reshape(-1, 1, 2))] * 2
    hom = np.tile(np.identity(3), (3, 1, 1))
    want_d, want_v = clib.vertex_motion(W, H, R, C, 4, 4, feats, hom)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, feature_ellipse_row_count=4, feature_ellipse_col_count=4)
    got_d, got_v = s._vertex_motion_from_features(3, W, H, feats, hom)
    assert np.array_equal(got_v, want_v) and np.array_equal(got_d, want_d)
    assert np.abs(want_v).max() > 0


def test_no_pairs_and_bad_arguments(dev):
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    s = MeshFlowStabilizer(mesh_row_count=4, mesh_col_count=4)
    disp, vel = s._vertex_motion_from_features(1, 320, 240, [], np.identity(3).reshape(1, 3, 3))
    assert disp.shape == (1, 5, 5, 2) and not disp.any() and vel.shape == (0, 5, 5, 2)
    with pytest.raises(ValueError):
        s._vertex_motion_from_features(3, 320, 240, [(None, None)], np.tile(np.identity(3), (3, 1, 1)))
    with pytest.raises(ValueError):
        s._vertex_motion_from_features(2, 320, 240, [(np.zeros((3, 1, 2)), np.zeros((4, 1, 2)))], np.tile(np.identity(3), (2, 1, 1)))


def test_features_to_stabilized_frames_end_to_end(dev):
    """features -> displacements -> Jacobi -> warp, all on the device path, against the oracle chain."""
    from oracle import clib, gen_golden, meshflow_oracle as mo
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    W, H, R, C, F = 320, 240, 8, 8, 12
    feats, hom = gen_golden.motion_inputs(W, H, R, C, F, (150, 220), 31)
    frames = synthetic.frames_numpy(F, H, W, seed=31)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, feature_ellipse_row_count=5, feature_ellipse_col_count=5,
                           temporal_smoothing_radius=4, optimization_num_iterations=20)
    disp, hom2 = s._get_unstabilized_vertex_displacements_from_features(F, W, H, feats, hom)
    out, bounds, stab, score = s.stabilize_clip(list(frames), disp, hom2)
    want_disp, _ = clib.vertex_motion(W, H, R, C, 5, 5, feats, hom)
    assert np.array_equal(disp, want_disp)
    taps, lam, on = mo.jacobi_band_coefficients(F, W, H, 0, hom, 4)
    want_stab = clib.jacobi_banded(want_disp.reshape(F, -1), taps, lam, np.reciprocal(on), 4, 20).reshape(want_disp.shape)
    assert np.abs(stab - want_stab).max() <= 1e-9          # coefficient set-up differs in the last bits (eigvals)
    want_out, crop, bad = clib.warp_clip(frames, R, C, want_disp, stab)
    assert bad == 0 and np.array_equal(np.stack(out), want_out)


def test_randomised_motion_campaign(dev):
    """tests/fuzz_motion.py: random frame sizes, meshes up to 39x39, ellipses up to 3x the mesh, 0-5000 features per
    pair incl. points outside the frame, points exactly on vertices / ellipse extremes, repeated residuals."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_motion
    assert fuzz_motion.run(150, 7) == 0


def test_raw_c_abi_as_in_integration_md(dev, golden_dir):
    """The device-pointer ABI driven with nothing but ctypes (mf_malloc / mf_memcpy_* / mf_vertex_motion_f64), the way
    INTEGRATION.md shows a maintainer of the reference would bind it."""
    import ctypes
    from meshflow_amd import _lib
    lib = _lib.lib
    g, feats = _load_case(golden_dir, 'motion_small')
    early = np.ascontiguousarray(np.concatenate([e.reshape(-1, 2) for e, _ in feats]), np.float64)
    late = np.ascontiguousarray(np.concatenate([l.reshape(-1, 2) for _, l in feats]), np.float64)
    counts = [len(e) for e, _ in feats]
    offsets = np.cumsum([0] + counts).astype(np.int32)
    P, K, R, C = len(counts), len(early), int(g['R']), int(g['C'])
    hom = np.ascontiguousarray(g['hom'][:P], np.float64)
    vel = np.empty((P, R + 1, C + 1, 2), np.float32)
    disp = np.empty((P + 1, R + 1, C + 1, 2), np.float64)
    status = np.zeros(1, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    arrays = dict(early=early, late=late, offsets=offsets, hom=hom, vel=vel, disp=disp, status=status)
    d = {}
    for name, arr in arrays.items():
        d[name] = ctypes.c_void_p()
        _lib.check(lib.mf_malloc(ctypes.byref(d[name]), arr.nbytes))
    work = ctypes.c_void_p()
    _lib.check(lib.mf_malloc(ctypes.byref(work), lib.mf_vertex_motion_workspace_bytes(K, max(counts), P, R, C)))
    for name in ('early', 'late', 'offsets', 'hom', 'status'):
        _lib.check(lib.mf_memcpy_h2d(d[name], p(arrays[name]), arrays[name].nbytes, None))
    _lib.check(lib.mf_vertex_motion_f64(d['early'], d['late'], d['offsets'], d['hom'], P, K, max(counts), int(g['width']),
                                        int(g['height']), R, C, int(g['ell_rows']), int(g['ell_cols']), d['vel'], d['disp'],
                                        work, d['status'], None))
    for name in ('vel', 'disp', 'status'):
        _lib.check(lib.mf_memcpy_d2h(p(arrays[name]), d[name], arrays[name].nbytes, None))
    _lib.check(lib.mf_stream_synchronize(None))
    for ptr in list(d.values()) + [work]:
        _lib.check(lib.mf_free(ptr))
    assert status[0] == 0
    assert np.array_equal(vel, g['velocities']) and np.array_equal(disp, g['displacements'])


def test_extreme_feature_counts_vs_c_oracle(dev):
    """Pairs with 0, 1, 2, 4,096 / 4,097, 65,537 and 120,000 features (half of them piled onto one spot: thousands of values under one
    vertex's median), ellipses larger than the frame, a 64 x 64 mesh: bit-identical to the C oracle (mfs.py:316-452 has no limits)."""
    from oracle import clib
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    rng = np.random.default_rng(12)
    cases = [(640, 360, 4, 4, 3, 3, [30000, 1, 0, 2, 65537]), (64, 48, 1, 1, 1, 1, [5, 4096, 4097]), (1920, 1080, 16, 16, 10, 10, [120000]),
             (33, 17, 2, 3, 40, 40, [7, 300]), (320, 240, 64, 64, 1, 1, [2000, 2000])]
    for W, H, R, C, er, ec, counts in cases:
        feats, F = [], len(counts) + 1
        for k in counts:
            if k == 0:
                feats.append((None, None))
                continue
            e = np.stack([rng.uniform(-5, W + 5, k), rng.uniform(-5, H + 5, k)], -1)
            if k >= 30000:
                e[: k // 2] = np.array([W * 0.4, H * 0.6]) + rng.normal(0, 1.5, size=(k // 2, 2))
            late = e + rng.normal(0, 2.0, size=e.shape) + np.array([1.5, -0.5])
            feats.append((e.reshape(-1, 1, 2), late.reshape(-1, 1, 2)))
        hom = np.tile(np.identity(3), (F, 1, 1))
        hom[:-1, :2, 2] = rng.normal(0, 1.0, size=(F - 1, 2))
        want_d, want_v = clib.vertex_motion(W, H, R, C, er, ec, feats, hom, openmp=True)
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, feature_ellipse_row_count=er, feature_ellipse_col_count=ec)
        got_d, got_v = s._vertex_motion_from_features(F, W, H, feats, hom)
        assert np.array_equal(got_v, want_v) and np.array_equal(got_d, want_d), (W, H, R, C, er, ec, counts)

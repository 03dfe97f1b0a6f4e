"""oracle/motion_oracle.py against the reference's own outputs (tests/golden/motion_*.npz, written by
oracle/gen_golden.py: the reference's mfs.py:236-452 executed with synthetic features) and known answers."""
import os

import numpy as np
import pytest

from oracle import motion_oracle as mt

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ('motion_small', 'motion_1080p', 'motion_ragged')


def load_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    off = g['offsets']
    feats = [(g['early'][off[t]:off[t + 1]].reshape(-1, 1, 2), g['late'][off[t]:off[t + 1]].reshape(-1, 1, 2))
             for t in range(len(off) - 1)]
    geo = tuple(int(g[k]) for k in ('width', 'height', 'R', 'C', 'ell_rows', 'ell_cols'))
    return g, feats, geo


@pytest.mark.parametrize('name', CASES)
def test_inputs_regenerate_from_the_hash_generator(name):
    """The committed feature arrays are what meshflow_amd.synthetic produces today (so they never drift)."""
    from oracle import gen_golden
    g, feats, (W, H, R, C, er, ec) = load_case(name)
    f2, hom = gen_golden.motion_inputs(W, H, R, C, int(g['F']), tuple(int(v) for v in g['per_pair']), int(g['seed']))
    assert np.array_equal(hom, g['hom'])
    for (e, l), (e2, l2) in zip(feats, f2):
        assert np.array_equal(e, e2) and np.array_equal(l, l2)


@pytest.mark.parametrize('name', CASES)
def test_splat_lists_match_reference(name):
    g, feats, (W, H, R, C, er, ec) = load_case(name)
    lx, ly = mt.vertex_nearby_feature_residual_velocities(W, H, R, C, er, ec, feats[0][0], feats[0][1], g['hom'][0])
    counts = np.array([[len(v) for v in row] for row in lx], dtype=np.int32)
    assert np.array_equal(counts, g['counts0'])
    assert np.array_equal(np.array([x for row in lx for v in row for x in v]), g['lists0_x'])
    assert np.array_equal(np.array([y for row in ly for v in row for y in v]), g['lists0_y'])


@pytest.mark.parametrize('name', CASES)
def test_velocities_and_displacements_match_reference(name):
    g, feats, (W, H, R, C, er, ec) = load_case(name)
    disp, vel = mt.unstabilized_vertex_displacements(W, H, R, C, er, ec, feats, g['hom'])
    assert vel.dtype == np.float32 and disp.dtype == np.float64
    assert np.array_equal(vel, g['velocities'])
    assert np.array_equal(disp, g['displacements'])


def test_ragged_case_has_uncovered_vertices():
    g, _, _ = load_case('motion_ragged')
    assert (g['counts0'] == 0).any() and (g['counts0'] > 0).any()


def test_median_or_zero():
    assert mt.median_or_zero([]) == 0.0
    assert mt.median_or_zero([3.0]) == 3.0
    assert mt.median_or_zero([3.0, 1.0]) == 2.0
    assert mt.median_or_zero([5.0, 1.0, 3.0]) == 3.0
    assert mt.median_or_zero([4.0, 1.0, 3.0, 2.0]) == 2.5
    a, b = 0.1, 0.30000000000000004
    assert mt.median_or_zero([b, a]) == (a + b) / 2


def test_median_blur3_known_answers():
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    out = mt.median_blur3_f32(img)
    # corner (0,0): neighbourhood with replicated borders = {0,0,1, 0,0,1, 4,4,5} -> sorted median = 1
    assert out[0, 0] == 1.0
    # interior (1,1): {0,1,2,4,5,6,8,9,10} -> 5
    assert out[1, 1] == 5.0
    # bottom-right (2,3): {6,7,7, 10,11,11, 10,11,11} -> 10
    assert out[2, 3] == 10.0
    spike = np.zeros((5, 5), dtype=np.float32)
    spike[2, 2] = 100.0
    assert not mt.median_blur3_f32(spike).any()


def test_ellipse_cover_known_answers():
    # feature at the centre of a 1600x800 frame with a 16x8 mesh -> (row 4.0, col 8.0); ellipse 4 rows x 6 cols
    spans = mt.ellipse_cover(800.0, 400.0, 1600, 800, 8, 16, 4, 6)
    # rows 2..6; half-widths: 6*sqrt(1/4 - (d/4)^2) = 0 (d=2), 2.598 (d=1), 3 (d=0)
    assert spans == [(2, 8, 8), (3, 6, 10), (4, 5, 11), (5, 6, 10), (6, 8, 8)]
    # clipped at the frame corner
    spans = mt.ellipse_cover(0.0, 0.0, 1600, 800, 8, 16, 4, 6)
    assert spans == [(0, 0, 3), (1, 0, 2), (2, 0, 0)]
    # a feature outside the mesh rows covers nothing
    assert mt.ellipse_cover(800.0, -900.0, 1600, 800, 8, 16, 4, 6) == []


def test_perspective_transform_f64_degenerate_w():
    H = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 0.0]])
    out = mt.perspective_transform_f64(np.array([[3.0, 4.0]]), H)
    assert np.array_equal(out, [[0.0, 0.0]])
    H = np.array([[2, 0, 1], [0, 3, -1], [0, 0, 2.0]])
    assert np.array_equal(mt.perspective_transform_f64(np.array([[3.0, 4.0]]), H), [[3.5, 5.5]])


# ---- the C restatement (oracle/motion_oracle.c) against the goldens and the Python restatement ----

@pytest.mark.parametrize('name', CASES)
def test_c_oracle_matches_reference(name):
    from oracle import clib
    g, feats, (W, H, R, C, er, ec) = load_case(name)
    for omp in (False, True):
        disp, vel = clib.vertex_motion(W, H, R, C, er, ec, feats, g['hom'], openmp=omp)
        assert np.array_equal(vel, g['velocities'])
        assert np.array_equal(disp, g['displacements'])


def test_c_oracle_matches_python_on_odd_geometry():
    from oracle import clib, gen_golden
    for (W, H, R, C, er, ec, F, per_pair, seed) in ((321, 243, 5, 9, 7, 3, 4, (40, 90), 11),
                                                  (640, 480, 12, 7, 2, 9, 3, (200, 260), 12),
                                                  (200, 100, 1, 1, 10, 10, 3, (5, 9), 13)):
        feats, hom = gen_golden.motion_inputs(W, H, R, C, F, per_pair, seed)
        feats[1] = (None, None)                                     # a pair without features: global motion only
        d0, v0 = mt.unstabilized_vertex_displacements(W, H, R, C, er, ec, feats, hom)
        d1, v1 = clib.vertex_motion(W, H, R, C, er, ec, feats, hom)
        assert np.array_equal(v0, v1) and np.array_equal(d0, d1)

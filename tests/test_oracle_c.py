"""The C oracle (oracle/warp_oracle.c) against the NumPy oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

from meshflow_amd import synthetic
from oracle import clib, meshflow_oracle as mo


@pytest.mark.parametrize('H,W,R,C,seed,kw', [
    (64, 96, 4, 4, 3, dict(jitter_sigma=1.0)),
    (50, 70, 3, 5, 4, dict(jitter_sigma=0.5)),
    (40, 130, 2, 9, 5, dict(translation_sigma=6.0)),
])
def test_c_warp_is_bit_identical_to_numpy_painter_loop(H, W, R, C, seed, kw):
    frames, disp, hom = synthetic.clip(4, H, W, R, C, seed=seed, kind='noise', **kw)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 10)
    for f in range(1, 4):
        out, crop, mx, my = mo.warp_frame(frames[f], R, C, disp[f], stab[f])
        table, bad = clib.cell_table(W, H, R, C, disp[f], stab[f])
        assert bad == 0
        for k, (Hf, Hi, rect) in enumerate(mo.cell_tables(W, H, R, C, disp[f], stab[f])):
            np.testing.assert_array_equal(mo.invert3x3(Hf).reshape(9), table[k, clib.OFF_M:clib.OFF_M + 9])
            np.testing.assert_array_equal(Hi.reshape(9), table[k, clib.OFF_HI:clib.OFF_HI + 9])
            assert tuple(table[k, clib.OFF_RECT:clib.OFF_RECT + 4]) == rect
        o2, c2, mx2, my2 = clib.warp_frame(frames[f], R, C, table, want_maps=True)
        np.testing.assert_array_equal(o2, out)
        assert tuple(c2) == crop
        np.testing.assert_array_equal(mx2, mx.astype(np.float32))
        np.testing.assert_array_equal(my2, my.astype(np.float32))
        o3, c3 = clib.warp_frame(frames[f], R, C, table, use_bbox=True)       # culled variant == brute force
        np.testing.assert_array_equal(o3, out)
        np.testing.assert_array_equal(c3, c2)


def test_bbox_culling_equals_brute_force_on_a_larger_frame():
    H, W, R, C = 180, 320, 16, 16
    frames, disp, hom = synthetic.clip(3, H, W, R, C, seed=8, kind='noise', translation_sigma=5.0, jitter_sigma=1.0)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 10)
    a, ca, _ = clib.warp_clip(frames, R, C, disp, stab, use_bbox=False)
    b, cb, _ = clib.warp_clip(frames, R, C, disp, stab, use_bbox=True, openmp=True)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ca, cb)
    # every pixel the brute force assigns to a cell lies inside that cell's recorded box
    table, _ = clib.cell_table(W, H, R, C, disp[2], stab[2])
    _, _, mx, my = clib.warp_frame(frames[2], R, C, table, want_maps=True)
    covered = mx != np.float32(W + 1)
    assert covered.mean() > 0.9


def test_c_jacobi_matches_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, 'jacobi_cfg2_subset.npz'))
    F, omega, iters = int(g['F']), int(g['omega']), int(g['iters'])
    taps, lam, on = mo.jacobi_band_coefficients(F, 1920, 1080, 0, g['hom'], omega)
    x = clib.jacobi_banded(g['inputs'].reshape(F, -1), taps, lam, np.reciprocal(on), omega, iters)
    assert np.abs(x.reshape(g['outputs'].shape) - g['outputs']).max() < 1e-10 * max(1.0, np.abs(g['outputs']).max())
    y = clib.jacobi_banded(g['inputs'].reshape(F, -1), taps, lam, np.reciprocal(on), omega, iters, openmp=True)
    np.testing.assert_array_equal(x, y)


def test_c_find_homography_equals_numpy():
    g = np.random.default_rng(5)
    for _ in range(50):
        src = np.array([[0, 0], [60, 0], [0, 34], [60, 34]], float) + [g.integers(0, 1800), g.integers(0, 1000)]
        dst = src + g.normal(0, 2, (4, 2))
        np.testing.assert_array_equal(clib.find_homography_4pt(src, dst), mo.find_homography_4pt(src, dst))
    assert clib.find_homography_4pt(np.zeros((4, 2)), np.ones((4, 2))) is None


def test_degenerate_cells_are_counted():
    z = np.zeros((5, 5, 2))
    s = z.copy()
    gx = np.array([np.ceil(95 * c / 4) for c in range(5)])
    s[:, :, 0] = -gx[None, :]
    table, bad = clib.cell_table(96, 64, 4, 4, z, s)
    assert bad == 16 and (table[:, clib.OFF_STATUS] == 1).all()

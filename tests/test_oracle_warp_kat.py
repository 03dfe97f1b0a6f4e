"""Known-answer tests that pin the warp half of the oracle (there is no cv2 here to compare with).

Each test states a fact that holds for OpenCV's remap / warpPerspective / perspectiveTransform /
findHomography semantics independent of their implementation details."""
import numpy as np
import pytest

from meshflow_amd import synthetic
from oracle import clib, meshflow_oracle as mo

H, W, R, C = 64, 96, 4, 4


@pytest.fixture(scope='module')
def frames():
    return synthetic.frames_numpy(2, H, W, seed=1, kind='noise')


def _zero():
    return np.zeros((R + 1, C + 1, 2))


def test_identity_motion_reproduces_the_frame(frames):
    out, crop, mx, my = mo.warp_frame(frames[0], R, C, _zero(), _zero())
    np.testing.assert_array_equal(out, frames[0])
    assert crop == (0, 0, W - 1, H - 1)
    np.testing.assert_array_equal(mx, np.broadcast_to(np.arange(W, dtype=np.float64), (H, W)))
    np.testing.assert_array_equal(my, np.broadcast_to(np.arange(H, dtype=np.float64)[:, None], (H, W)))


def test_integer_translation_shifts_and_fills_border(frames):
    s = _zero(); s[..., 0] = 5; s[..., 1] = -3
    out, crop, _, _ = mo.warp_frame(frames[0], R, C, _zero(), s)
    want = np.empty_like(frames[0]); want[...] = (0, 0, 255)         # default border colour, BGR (mfs.py:48)
    want[0:H - 3, 5:W] = frames[0][3:H, 0:W - 5]
    np.testing.assert_array_equal(out, want)
    assert crop == (5, 0, W - 1, H - 4)          # left = 5 (maps to x_u = 0), bottom = H-4 (maps to y_u = H-1)


def test_half_pixel_translation_is_the_fixed_point_average(frames):
    s = _zero(); s[..., 0] = 0.5
    out, _, _, _ = mo.warp_frame(frames[0], R, C, _zero(), s)
    a = frames[0][:, 0:W - 1].astype(np.int64); b = frames[0][:, 1:W].astype(np.int64)
    np.testing.assert_array_equal(out[:, 1:W], ((16384 * a + 16384 * b + (1 << 14)) >> 15).astype(np.uint8))
    # column 0 samples x = -0.5: half border colour, half pixel 0
    bc = np.array([0, 0, 255], dtype=np.int64)
    np.testing.assert_array_equal(out[:, 0], ((16384 * bc + 16384 * frames[0][:, 0].astype(np.int64) + (1 << 14)) >> 15).astype(np.uint8))


def test_custom_border_colour(frames):
    s = _zero(); s[..., 1] = 7
    out, _, _, _ = mo.warp_frame(frames[0], R, C, _zero(), s, border_bgr=(12, 34, 56))
    assert (out[:7] == (12, 34, 56)).all()


def test_single_vertex_perturbation_touches_only_its_four_cells(frames):
    s = _zero(); s[2, 2] = (1.5, -1.25)
    out, _, mx, my = mo.warp_frame(frames[0], R, C, _zero(), s)
    gx = [int(np.ceil((W - 1) * c / C)) for c in range(C + 1)]
    gy = [int(np.ceil((H - 1) * r / R)) for r in range(R + 1)]
    changed = (mx != np.arange(W)[None, :]) | (my != np.arange(H)[:, None])
    ys, xs = np.nonzero(changed)
    assert xs.min() >= gx[1] - 1 and xs.max() <= gx[3] + 1 and ys.min() >= gy[1] - 1 and ys.max() <= gy[3] + 1
    assert changed.any()


def test_last_cell_in_row_major_order_owns_the_overlap():
    """Neighbouring cells' masks overlap by about a pixel; the later cell's coordinates must win."""
    frames, disp, hom = synthetic.clip(3, H, W, R, C, seed=9, kind='noise', jitter_sigma=1.5)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 10)
    f = 2
    _, _, mx, my = mo.warp_frame(frames[f], R, C, disp[f], stab[f])
    cells = mo.cell_tables(W, H, R, C, disp[f], stab[f])
    xy = np.swapaxes(np.indices((W, H), dtype=np.float32), 0, 2)
    owner = np.full((H, W), -1)
    for k, (Hf, Hi, rect) in enumerate(cells):
        owner[mo.warp_perspective_rect_mask(rect, Hf, W, H)] = k
    overlaps = 0
    for k, (Hf, Hi, rect) in enumerate(cells):
        m = mo.warp_perspective_rect_mask(rect, Hf, W, H)
        overlaps += int((m & (owner != k)).sum())
        pts = mo.perspective_transform_f32(xy, Hi)
        sel = owner == k
        np.testing.assert_array_equal(mx[sel], pts[..., 0][sel].astype(np.float64))
        np.testing.assert_array_equal(my[sel], pts[..., 1][sel].astype(np.float64))
    assert overlaps > 50         # the overlap band exists, so the ordering matters


def test_rect_mask_is_the_nonzero_pattern_of_a_full_bilinear_warp():
    g = np.random.default_rng(3)
    for _ in range(4):
        L, T = int(g.integers(2, 20)), int(g.integers(2, 12))
        Rt, B = L + int(g.integers(3, 15)), T + int(g.integers(3, 10))
        src = np.array([[L, T], [Rt, T], [L, B], [Rt, B]], dtype=np.float64)
        Hf = mo.find_homography_4pt(src, src + g.normal(0, 1.2, (4, 2)))
        mask_img = np.zeros((32, 48)); mask_img[T:B + 1, L:Rt + 1] = 255
        full = mo.warp_perspective_f64_bilinear(mask_img, Hf, 48, 32)
        np.testing.assert_array_equal(full != 0, mo.warp_perspective_rect_mask((L, T, Rt, B), Hf, 48, 32))


def test_find_homography_maps_the_four_points_and_matches_the_eigen_route():
    g = np.random.default_rng(0)
    worst = 0.0
    for _ in range(100):
        src = np.array([[0, 0], [120, 0], [0, 68], [120, 68]], float) + [g.integers(0, 1800), g.integers(0, 1000)]
        dst = src + g.normal(0, 3, (4, 2))
        Hg = mo.find_homography_4pt(src, dst)                 # closed form (what the C oracle and the HIP kernel compute)
        He = mo.find_homography_4pt(src, dst, 'eigh')
        He = He * np.sign(He[2, 2]) * np.sign(Hg[2, 2])
        worst = max(worst, np.abs(Hg - He).max() / np.abs(Hg).max())
        Hx = mo.find_homography_4pt(src, dst, 'gauss')        # rounds 1-3: 8 x 8 elimination with partial pivoting
        worst = max(worst, np.abs(Hg - Hx).max() / np.abs(Hg).max())
        p = np.c_[src.astype(np.float32).astype(np.float64), np.ones(4)] @ Hg.T
        np.testing.assert_allclose(p[:, :2] / p[:, 2:], dst.astype(np.float32).astype(np.float64), atol=1e-7)
        assert abs(Hg[2, 2] - 1.0) < 1e-15
    assert worst < 1e-9          # closed form vs Gaussian elimination vs smallest eigenvector of L^T L: the same solution
    assert mo.find_homography_4pt([[0, 0], [0, 0], [0, 0], [0, 0]], [[0, 0], [1, 0], [0, 1], [1, 1]]) is None


@pytest.mark.parametrize('triple', [(0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3)])
def test_find_homography_refuses_any_three_collinear_corners(triple):
    """a quad with ANY three corners on a line has no homography (the 8 equations are rank deficient) -- also when the
    triple includes the TL corner, which the closed form's own denominator does not look at.  NumPy and C oracle agree, for a
    degenerate destination and for a degenerate source."""
    rect = np.array([[10, 20], [130, 20], [10, 88], [130, 88]], dtype=np.float32)          # TL, TR, BL, BR
    a, b, c = triple
    bad = rect.copy()
    bad[c] = bad[a] + 0.5 * (bad[b] - bad[a])                    # the third corner onto the line through the other two (exact in float32)
    assert mo.find_homography_4pt(rect, bad) is None and mo.find_homography_4pt(bad, rect) is None
    assert clib.find_homography_4pt(rect, bad) is None and clib.find_homography_4pt(bad, rect) is None
    # one float32 ulp off the line is a (wild but well-defined) homography again, as it is for cv2.findHomography: only a system without
    # a unique solution is refused
    off = bad.copy()
    off[c, 1] = np.nextafter(off[c, 1], np.float32(1e9)) if bad[a, 0] != bad[b, 0] else off[c, 1]
    off[c, 0] = np.nextafter(off[c, 0], np.float32(1e9)) if bad[a, 0] == bad[b, 0] else off[c, 0]
    assert mo.find_homography_4pt(rect, off) is not None and clib.find_homography_4pt(rect, off) is not None
    good = rect + np.array([[1.5, -0.5], [-2, 1], [0.5, 2], [3, -1]], dtype=np.float32)     # an ordinary quad still solves
    assert mo.find_homography_4pt(rect, good) is not None and clib.find_homography_4pt(rect, good) is not None
    # (the elimination solver kept as a cross-check returns the SINGULAR matrix of such a system -- what cv2.findHomography would hand to
    # cv2.warpPerspective, whose inverse is then the zero matrix: the closed form and the kernel report the cell as degenerate instead)
    sing = mo.find_homography_4pt(rect, np.array([[10, 20], [70, 20], [130, 20], [130, 88]], np.float32), solver='gauss')
    assert sing is None or abs(np.linalg.det(sing)) < 1e-9


def test_invert3x3_closed_form():
    g = np.random.default_rng(1)
    for _ in range(20):
        A = np.eye(3) + 0.1 * g.normal(size=(3, 3))
        np.testing.assert_allclose(mo.invert3x3(A) @ A, np.eye(3), atol=1e-13)
        np.testing.assert_array_equal(mo.invert3x3(A), clib.invert3x3(A))
    assert not mo.invert3x3(np.zeros((3, 3))).any()


def test_remap_border_rules():
    src = synthetic.frames_numpy(1, 6, 8, seed=2, kind='noise')[0]
    mx = np.array([[-1.0, -0.5, 7.0, 7.5, 8.0, 9.0, 3.25]], dtype=np.float32)
    my = np.array([[2.0, 2.0, 5.0, 5.5, 2.0, 7.0, 1.75]], dtype=np.float32)
    out = mo.remap_bilinear_u8c3(src, mx, my, (0, 0, 255))[0].astype(np.int64)
    bc = np.array([0, 0, 255])
    s = src.astype(np.int64)
    # x = -1: ix = -1, fx = 0 -> weight entirely on the outside tap (-1, 2): border colour
    np.testing.assert_array_equal(out[0], bc)
    np.testing.assert_array_equal(out[1], (16 * 32 * 32 * bc + 16 * 32 * 32 * s[2, 0] + 16384) >> 15)
    np.testing.assert_array_equal(out[2], s[5, 7])                                  # last pixel, weights on (7,5) only
    np.testing.assert_array_equal(out[3], (256 * 32 * s[5, 7] + 3 * 256 * 32 * bc + 16384) >> 15)
    np.testing.assert_array_equal(out[4], bc)                                       # ix = W: wholly outside
    np.testing.assert_array_equal(out[5], bc)
    w00, w01, w10, w11 = 24 * 8, 8 * 8, 24 * 24, 8 * 24
    np.testing.assert_array_equal(out[6], ((w00 * s[1, 3] + w01 * s[1, 4] + w10 * s[2, 3] + w11 * s[2, 4]) * 32 + 16384) >> 15)


def test_crop_scan_uses_strict_less_than_one():
    s = _zero(); s[..., 0] = 1.0     # u = x - 1: |u| < 1 only at x = 1 (u = 0), since u(0) = -1, u(2) = 1
    fr = synthetic.frames_numpy(1, H, W, seed=3)[0]
    _, crop, mx, _ = mo.warp_frame(fr, R, C, _zero(), s)
    assert crop[0] == 1 and crop[2] == W - 1

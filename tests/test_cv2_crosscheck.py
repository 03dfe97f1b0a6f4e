"""Optional cross-check of the oracle's OpenCV restatements against a real OpenCV -- the ONE run that settles "parity unpinned".

OpenCV is not installed in the build image nor on the GPU box, so these tests normally SKIP; wherever `cv2`
is importable they pin the restated calls of the warp path, cv2.resize and the two calls of the vertex-motion row
(perspectiveTransform on float64 points, medianBlur) against the real thing.  `tools/settle_parity.sh` is the one command to run on
such a box; it prints (and writes, $MESHFLOW_PARITY_REPORT) a JSON verdict.

WHICH OpenCV is modelled.  The oracle restates the FIXED-POINT 8-bit kernels: `remap` / `warpPerspective` with INTER_BITS = 5
coordinates and the 2^15-weight table (imgwarp.cpp), `resize` with 11-bit coefficients (resize.cpp), the 4-point `findHomography` through
the normalised DLT (fundam.cpp).  Those are OpenCV 4.5 ... 4.10 (the reference's era -- numpy 1.22 / tqdm 4.56 in its requirements --
is 4.5.x; 3.4 / 4.0-4.4 share the kernels but were never the target).  OpenCV >= 4.11 replaced the linear `remap` / `warpAffine` /
`warpPerspective` kernels for 8-bit images by ones that blend with float weights: against such a build the pixel results may differ in
the last bit, which is a VERSION difference, not a parity failure.  Therefore:
  * inside the modelled range every comparison is BIT-EXACT (up to the stated rounding ties of the two homography solvers);
  * outside it the assertions are BASELINE.json's bars -- <= 1 LSB on uint8 pixels, 1e-4 on coordinates -- and the number of
    values that are not bit-equal is reported as information (`mismatching` in the report), never asserted."""
import atexit
import json
import os
import re
import sys

import numpy as np
import pytest

try:
    import cv2
except ImportError:                                     # no OpenCV on this box (the build image, the GPU box): every test here is skipped --
    cv2 = None                                          # as a MARK, so that `-m gpu` deselects them instead of reporting a skip
pytestmark = pytest.mark.skipif(cv2 is None, reason='no OpenCV on this box (tools/settle_parity.sh is for a box that has one)')

from meshflow_amd import synthetic                      # noqa: E402
from oracle import meshflow_oracle as mo                # noqa: E402

MODELLED_RANGE = ((4, 5, 0), (4, 11, 0))                # [first modelled release, first release with the float-weight kernels)


def _cv2_version():
    v = getattr(cv2, '__version__', None)
    if v is None:                                       # the stand-in of tests/test_cv2_crosscheck_bitrot.py: the oracle itself
        return None
    m = re.match(r'(\d+)\.(\d+)(?:\.(\d+))?', v)
    return (int(m.group(1)), int(m.group(2)), int(m.group(3) or 0)) if m else (0, 0, 0)


CV2_VERSION = _cv2_version() if cv2 is not None else (0, 0, 0)
BIT_EXACT = CV2_VERSION is None or MODELLED_RANGE[0] <= CV2_VERSION < MODELLED_RANGE[1]
_REPORT = {'cv2_version': getattr(cv2, '__version__', 'stand-in (the oracle itself: proves nothing about OpenCV)'),
           'modelled_range': 'OpenCV %d.%d.%d <= version < %d.%d.%d (fixed-point remap / warpPerspective / resize)' % (MODELLED_RANGE[0] + MODELLED_RANGE[1]),
           'assertion_mode': 'bit-exact' if BIT_EXACT else 'BASELINE.json bars (<= 1 LSB pixels, 1e-4 coordinates); mismatch counts are information',
           'checks': []}
if cv2 is not None:
    print(f'[cv2 cross-check] cv2 {_REPORT["cv2_version"]}; modelled: {_REPORT["modelled_range"]}; asserting {_REPORT["assertion_mode"]}', file=sys.stderr)


def _write_report():
    path = os.environ.get('MESHFLOW_PARITY_REPORT')
    if path and cv2 is not None and CV2_VERSION is not None:
        with open(path, 'w') as fh:
            json.dump(_REPORT, fh, indent=1)


atexit.register(_write_report)


def check_pixels(got, ref, what, ties=0):
    """uint8 (or boolean mask) results: bit-exact up to `ties` values inside the modelled range, <= 1 LSB outside it."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, what
    diff = np.abs(got.astype(np.int64) - ref.astype(np.int64))
    mism = int((diff != 0).sum())
    _REPORT['checks'].append({'check': what, 'kind': 'pixels', 'values': int(got.size), 'mismatching': mism, 'max_abs': int(diff.max()) if diff.size else 0})
    if BIT_EXACT:
        assert mism <= ties, f'{what}: {mism} of {got.size} values differ (cv2 {_REPORT["cv2_version"]} is inside the modelled range)'
    else:
        assert diff.max() <= 1, f'{what}: max |difference| {int(diff.max())} LSB (bar: 1)'


def check_coords(got, ref, what, exact=True, rtol=0.0, atol=0.0):
    """floating-point results: bit-exact (or within the stated solver tolerance) inside the modelled range, 1e-4 outside it."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref)
    mism = int((err != 0).sum())
    _REPORT['checks'].append({'check': what, 'kind': 'coordinates', 'values': int(got.size), 'mismatching': mism, 'max_abs': float(err.max()) if err.size else 0.0})
    if BIT_EXACT and exact:
        assert mism == 0, f'{what}: {mism} of {got.size} values differ'
    elif BIT_EXACT:
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=what)
    else:
        assert err.max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), f'{what}: max |difference| {err.max()}'


def test_find_homography_4pt():
    g = np.random.default_rng(0)
    for _ in range(50):
        src = (np.array([[0, 0], [120, 0], [0, 68], [120, 68]], float) + [g.integers(0, 1800), g.integers(0, 1000)]).astype(np.float32)
        dst = src.astype(np.float64) + g.normal(0, 3, (4, 2))
        ref, _ = cv2.findHomography(src, dst)
        got = mo.find_homography_4pt(src, dst)
        check_coords(got, ref, 'findHomography 4 points (mfs.py:1041-1042)', exact=False, rtol=1e-8, atol=1e-9)


def test_perspective_transform():
    g = np.random.default_rng(1)
    H = np.eye(3) + 0.01 * g.normal(size=(3, 3)); H[2, :2] *= 1e-3
    xy = np.swapaxes(np.indices((64, 48), dtype=np.float32), 0, 2).reshape(-1, 1, 2)
    ref = cv2.perspectiveTransform(xy, H)
    check_coords(mo.perspective_transform_f32(xy, H), ref, 'perspectiveTransform float32 points (mfs.py:1054)')


def test_warp_perspective_mask_pattern():
    g = np.random.default_rng(2)
    for _ in range(5):
        L, T = int(g.integers(2, 20)), int(g.integers(2, 12)); Rt, B = L + int(g.integers(3, 15)), T + int(g.integers(3, 10))
        src = np.array([[L, T], [Rt, T], [L, B], [Rt, B]], dtype=np.float32)
        Hf, _ = cv2.findHomography(src, src.astype(np.float64) + g.normal(0, 1.2, (4, 2)))
        mask = np.zeros((32, 48)); mask[T:B + 1, L:Rt + 1] = 255
        ref = cv2.warpPerspective(mask, Hf, (48, 32)) != 0
        got = mo.warp_perspective_rect_mask((L, T, Rt, B), Hf, 48, 32)
        check_pixels(got, ref, 'warpPerspective mask pattern (mfs.py:1052)', ties=1)          # at most a rounding tie


def test_remap_bilinear_constant_border():
    src = synthetic.frames_numpy(1, 40, 60, seed=3, kind='noise')[0]
    g = np.random.default_rng(3)
    mx = (np.arange(60, dtype=np.float32)[None, :] + g.normal(0, 3, (40, 60))).astype(np.float32)
    my = (np.arange(40, dtype=np.float32)[:, None] + g.normal(0, 3, (40, 60))).astype(np.float32)
    ref = cv2.remap(src, mx.reshape(40, 60, 1), my.reshape(40, 60, 1), cv2.INTER_LINEAR, borderValue=(0, 0, 255))
    check_pixels(mo.remap_bilinear_u8c3(src, mx, my, (0, 0, 255)), ref, 'remap bilinear constant border (mfs.py:1063-1069)')


def test_resize_linear():
    src = synthetic.frames_numpy(1, 37, 53, seed=4, kind='noise')[0]
    for (w, h) in ((60, 40), (53, 37), (106, 74), (55, 38)):
        check_pixels(mo.resize_linear_u8(src, w, h), cv2.resize(src, (w, h)), f'resize to {w}x{h} (mfs.py:1150)')


def test_whole_warp_against_reference_loop():
    """The reference's own per-cell loop (mfs.py:1031-1069) run with the real cv2 on one small frame."""
    H, W, R, C = 48, 64, 4, 4
    frames, disp, hom = synthetic.clip(2, H, W, R, C, seed=5, kind='noise', jitter_sigma=1.0)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 10)
    got, _, _, _ = mo.warp_frame(frames[1], R, C, disp[1], stab[1])
    grid = mo.vertex_x_y(W, H, R, C)
    rc_u = grid.reshape(R + 1, C + 1, 2)
    rc_s = (grid + (stab[1] - disp[1]).reshape(-1, 1, 2)).reshape(R + 1, C + 1, 2)
    map_x = np.full((H, W), W + 1); map_y = np.full((H, W), H + 1)
    xy = np.swapaxes(np.indices((W, H), dtype=np.float32), 0, 2).reshape(-1, 1, 2)
    for r in range(R):
        for c in range(C):
            ub = rc_u[r:r + 2, c:c + 2].reshape(-1, 2); sb = rc_s[r:r + 2, c:c + 2].reshape(-1, 2)
            Hf, _ = cv2.findHomography(ub, sb); Hi, _ = cv2.findHomography(sb, ub)
            mask = np.zeros((H, W))
            mask[int(ub[:, 1].min()):int(ub[:, 1].max()) + 1, int(ub[:, 0].min()):int(ub[:, 0].max()) + 1] = 255
            m = cv2.warpPerspective(mask, Hf, (W, H))
            pts = cv2.perspectiveTransform(xy, Hi).reshape(H, W, 2)
            map_x = np.where(m, pts[..., 0], map_x); map_y = np.where(m, pts[..., 1], map_y)
    ref = cv2.remap(frames[1], map_x.reshape(H, W, 1).astype(np.float32), map_y.reshape(H, W, 1).astype(np.float32),
                    cv2.INTER_LINEAR, borderValue=(0, 0, 255))
    check_pixels(got, ref, 'whole warp of one frame against the reference loop (mfs.py:1031-1069)', ties=int(1e-3 * got.size))   # only rounding ties of the two homography solvers may differ


def test_perspective_transform_float64_points():
    """The CV_64F flavour the reference's feature arrays take (mfs.py:420, promoted at mfs.py:578)."""
    from oracle import motion_oracle as mt
    rng = np.random.default_rng(5)
    pts = rng.uniform(0, 1900, (500, 1, 2))
    H = np.identity(3) + rng.normal(0, 0.01, (3, 3))
    H[2, :2] = rng.normal(0, 1e-6, 2)
    H[2, 2] = 1.0
    check_coords(mt.perspective_transform_f64(pts, H), cv2.perspectiveTransform(pts, H), 'perspectiveTransform float64 points (mfs.py:420)')


def test_median_blur_3x3_float32():
    """cv2.medianBlur on the (R+1) x (C+1) float32 velocity planes (mfs.py:359-360): replicated borders."""
    from oracle import motion_oracle as mt
    rng = np.random.default_rng(6)
    for shape in ((17, 17), (33, 33), (4, 6), (2, 2)):
        img = rng.normal(0, 3, shape).astype(np.float32)
        check_coords(mt.median_blur3_f32(img), cv2.medianBlur(img, 3), f'medianBlur 3x3 float32 {shape} (mfs.py:359-360)')


def test_vertex_velocities_against_reference_code_with_real_cv2():
    """The reference's own _get_unstabilized_vertex_velocities with the real cv2 (needs /root/reference importable)."""
    ref_dir = os.environ.get('MESHFLOW_REFERENCE_DIR', '/root/reference')
    if not os.path.exists(os.path.join(ref_dir, 'meshflowstabilizer.py')):
        pytest.skip('reference not present')
    sys.path.insert(0, ref_dir)
    import meshflowstabilizer as mfs
    from oracle import gen_golden, motion_oracle as mt
    W, H, R, C, er, ec = 640, 360, 8, 8, 5, 5
    feats, hom = gen_golden.motion_inputs(W, H, R, C, 4, (120, 180), 3)
    s = mfs.MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, feature_ellipse_row_count=er, feature_ellipse_col_count=ec)
    frames = [np.zeros((H, W, 3), np.uint8) for _ in range(4)]
    index = {id(f): i for i, f in enumerate(frames)}
    s._get_matched_features_and_homography = lambda a, b: (*feats[index[id(a)]], hom[index[id(a)]])
    for t in range(3):
        want = s._get_unstabilized_vertex_velocities(frames[t], frames[t + 1])[0]
        got = mt.unstabilized_vertex_velocities(W, H, R, C, er, ec, feats[t][0], feats[t][1], hom[t])
        check_coords(got, want, 'vertex velocities against the reference code (mfs.py:287-452)')


# ---- config-2 geometry (1920x1080, 16x16 mesh) and the committed goldens: one run anywhere `cv2` exists settles rows a-7 / a-8 / a-9 /
# ---- a-11 / f-1 of SURVEY.md section 8 --------------------------------------------------------------------------------------------

def _cfg2_cells(n_cells=24, seed=11):
    """A sample of config-2 cells of one frame of the bench clip: (unstabilized float32 corners, stabilized float64 corners, rect)."""
    H, W, R, C = 1080, 1920, 16, 16
    disp, hom = synthetic.motion(40, R, C, seed=0)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 10, 100)
    grid = mo.vertex_x_y(W, H, R, C).reshape(R + 1, C + 1, 2)
    moved = grid.astype(np.float64) + (stab[20] - disp[20])
    g = np.random.default_rng(seed)
    cells = []
    for k in g.choice(R * C, n_cells, replace=False):
        r, c = divmod(int(k), C)
        ub = grid[r:r + 2, c:c + 2].reshape(-1, 2)
        sb = moved[r:r + 2, c:c + 2].reshape(-1, 2)
        cells.append((ub, sb, (int(ub[:, 0].min()), int(ub[:, 1].min()), int(ub[:, 0].max()), int(ub[:, 1].max()))))
    return H, W, cells


def test_cfg2_find_homography_both_directions():
    """mfs.py:1041-1042 on config-2 cells: the restated 4-point solver against cv2.findHomography, both directions."""
    _, _, cells = _cfg2_cells()
    for ub, sb, _ in cells:
        for src, dst in ((ub, sb), (sb, ub)):
            ref, _ = cv2.findHomography(src, dst)
            check_coords(mo.find_homography_4pt(src, dst), ref, 'findHomography on config-2 cells (mfs.py:1041-1042)', exact=False, rtol=1e-8, atol=1e-8)


def test_cfg2_warp_perspective_mask_and_perspective_transform():
    """mfs.py:1050-1054 at full frame size: the non-zero pattern of the warped cell mask and the float32 coordinates of every pixel."""
    H, W, cells = _cfg2_cells(n_cells=6)
    xy = np.swapaxes(np.indices((W, H), dtype=np.float32), 0, 2).reshape(-1, 1, 2)
    for ub, sb, (L, T, Rt, B) in cells:
        Hf, _ = cv2.findHomography(ub, sb)
        Hi, _ = cv2.findHomography(sb, ub)
        mask = np.zeros((H, W)); mask[T:B + 1, L:Rt + 1] = 255
        ref = cv2.warpPerspective(mask, Hf, (W, H)) != 0
        got = mo.warp_perspective_rect_mask((L, T, Rt, B), Hf, W, H)
        check_pixels(got, ref, 'warpPerspective mask pattern at 1920x1080 (mfs.py:1052)', ties=2)      # at most a rounding tie or two along 400 px of edge
        check_coords(mo.perspective_transform_f32(xy, Hi), cv2.perspectiveTransform(xy, Hi), 'perspectiveTransform of every 1080p pixel (mfs.py:1054)')


def test_cfg2_remap_full_frame():
    """mfs.py:1063-1069 on a 1920x1080 frame with a smooth few-pixel displacement field incl. samples beyond all four borders."""
    H, W = 1080, 1920
    src = synthetic.frames_numpy(1, H, W, seed=0, kind='noise')[0]
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    mx = (xx + 9.0 * np.sin(yy / 97.0) - 3.3).astype(np.float32)
    my = (yy + 7.0 * np.cos(xx / 131.0) + 2.7).astype(np.float32)
    ref = cv2.remap(src, mx.reshape(H, W, 1), my.reshape(H, W, 1), cv2.INTER_LINEAR, borderValue=(0, 0, 255))
    check_pixels(mo.remap_bilinear_u8c3(src, mx, my, (0, 0, 255)), ref, 'remap of a 1920x1080 frame (mfs.py:1063-1069)')


def test_cfg2_resize_of_the_real_crop_rectangle():
    """mfs.py:1150: cv2.resize of the bench clip's crop rectangle (13, 11, 1909, 1068) back to 1920x1080 -- a 1.2 % up-scale."""
    H, W = 1080, 1920
    frame = synthetic.frames_numpy(1, H, W, seed=0, kind='noise')[0]
    for (l, t, r, b) in ((13, 11, 1909, 1068), (0, 0, W - 1, H - 1), (100, 37, 1500, 1000)):
        crop = frame[t:b + 1, l:r + 1]
        check_pixels(mo.resize_linear_u8(crop, W, H), cv2.resize(crop, (W, H)), f'resize of crop rectangle {(l, t, r, b)} to 1920x1080 (mfs.py:1150)')
        check_pixels(mo.crop_frames([frame], (l, t, r, b))[0], cv2.resize(crop, (W, H)), f'_crop_frames {(l, t, r, b)} (mfs.py:1111-1157)')


@pytest.mark.parametrize('name', ['warp_small', 'warp_ragged', 'warp_jitter', 'warp_shift', 'warp_mesh16'])
def test_reference_with_real_cv2_reproduces_the_committed_goldens(name):
    """The reference's OWN _get_stabilized_frames_and_crop_boundaries (mfs.py:909-1108) with the REAL cv2 against tests/golden/warp_*.npz
    (made by the same method under the stub cv2 whose four calls are the oracle's restatements): equal crop bounds, and frames equal up to
    rounding ties of the two homography solvers.  Needs /root/reference (or MESHFLOW_REFERENCE_DIR) importable."""
    ref_dir = os.environ.get('MESHFLOW_REFERENCE_DIR', '/root/reference')
    if not os.path.exists(os.path.join(ref_dir, 'meshflowstabilizer.py')):
        pytest.skip('reference not present')
    sys.path.insert(0, ref_dir)
    import meshflowstabilizer as mfs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'))
    s = mfs.MeshFlowStabilizer(mesh_row_count=int(g['R']), mesh_col_count=int(g['C']),
                               color_outside_image_area_bgr=tuple(int(v) for v in g['border']))
    out, bounds = s._get_stabilized_frames_and_crop_boundaries(int(g['F']), list(g['frames']), g['unstab'], g['stab'])
    if BIT_EXACT:
        assert tuple(int(b) for b in bounds) == tuple(int(b) for b in g['bounds'])
    else:                                                    # (a last-bit coordinate can move a crop edge by a pixel)
        assert max(abs(int(a) - int(b)) for a, b in zip(bounds, g['bounds'])) <= 1
    check_pixels(np.stack(out), g['out'], f'reference warp with the real cv2 against golden {name} (mfs.py:909-1108)', ties=int(1e-4 * g['out'].size))

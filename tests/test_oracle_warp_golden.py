"""The oracles' warp half against the REFERENCE's own `_get_stabilized_frames_and_crop_boundaries` (mfs.py:909-1108).

`tests/golden/warp_*.npz` were written by oracle/gen_golden.py, which runs the real reference method under a stub
`cv2` whose four calls (findHomography, warpPerspective -- the full float64 bilinear warp of the mask --,
perspectiveTransform, remap) are the restatements of oracle/meshflow_oracle.py.  Everything around those calls is the
reference's own NumPy: map templates, painter order, dtype promotions, edge scans, clip-level reduction (rows a-6, a-10,
a-12, a-13 of SURVEY.md section 8).  Bit-exact, both oracles."""
import os

import numpy as np
import pytest

CASES = ['warp_small', 'warp_ragged', 'warp_jitter', 'warp_shift', 'warp_mesh16']


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    return g, int(g['R']), int(g['C']), tuple(int(v) for v in g['border'])


@pytest.mark.parametrize('name', CASES)
def test_inputs_regenerate_from_the_hash_generator(golden_dir, name):
    """The stored inputs are what oracle/gen_golden.py's `warp_inputs` makes today (so the fixtures can be rebuilt)."""
    from oracle import gen_golden
    g, R, C, _ = load(golden_dir, name)
    case = [c for c in gen_golden.WARP_CASES if c[0] == name][0]
    _, W, H, R2, C2, F, kind, _, motion_kw, shift = case
    frames, unstab, stab = gen_golden.warp_inputs(W, H, R2, C2, F, kind, motion_kw, shift, seed=int(g['seed']))
    assert (R, C) == (R2, C2)
    np.testing.assert_array_equal(frames, g['frames'])
    np.testing.assert_array_equal(unstab, g['unstab'])
    np.testing.assert_array_equal(stab, g['stab'])


@pytest.mark.parametrize('name', CASES)
def test_numpy_oracle_equals_the_reference(golden_dir, name):
    from oracle import meshflow_oracle as mo
    g, R, C, border = load(golden_dir, name)
    out, bounds, _ = mo.stabilized_frames_and_crop_boundaries(list(g['frames']), R, C, g['unstab'], g['stab'], border)
    np.testing.assert_array_equal(np.stack(out), g['out'])
    assert tuple(bounds) == tuple(int(v) for v in g['bounds'])


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('use_bbox', [False, True])
def test_c_oracle_equals_the_reference(golden_dir, name, use_bbox):
    from oracle import clib
    g, R, C, border = load(golden_dir, name)
    out, crop, bad = clib.warp_clip(g['frames'], R, C, g['unstab'], g['stab'], border, use_bbox=use_bbox)
    assert bad == 0
    np.testing.assert_array_equal(out, g['out'])
    bounds = (crop[:, 0].max(), crop[:, 1].max(), crop[:, 2].min(), crop[:, 3].min())        # mfs.py:1103-1106
    assert tuple(int(v) for v in bounds) == tuple(int(v) for v in g['bounds'])


def test_vectorised_warp_perspective_equals_the_per_pixel_statement():
    """`warp_perspective_f64_bilinear_np` (what the golden generator's stub cv2.warpPerspective runs) against the
    per-pixel loop it vectorises, on a mask image and on a random-valued image."""
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    W, H = 40, 28
    for t in range(3):
        g = synthetic.normal(np.arange(9) + 9 * t, seed=77).reshape(3, 3)
        Hf = np.identity(3) + g * np.array([[0.03, 0.03, 3.0], [0.03, 0.03, 3.0], [1e-4, 1e-4, 0.0]])
        mask = np.zeros((H, W)); mask[6:20, 9:31] = 255
        img = 255.0 * synthetic.uniform01(np.arange(H * W) + t, seed=5).reshape(H, W)
        for src in (mask, img):
            np.testing.assert_array_equal(mo.warp_perspective_f64_bilinear_np(src, Hf, W, H),
                                          mo.warp_perspective_f64_bilinear(src, Hf, W, H))

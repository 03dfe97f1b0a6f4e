"""The NumPy oracle against vectors produced by the reference itself (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from oracle import meshflow_oracle as mo


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('F,omega', [(12, 3), (48, 10), (300, 10), (300, 30)])
@pytest.mark.parametrize('definition', [0, 1, 2, 3])
def test_coefficients_match_reference(golden_dir, F, omega, definition):
    g = _load(golden_dir, 'coeffs.npz')
    key = f'F{F}_O{omega}_D{definition}'
    hom = g[key + '_hom']
    lam = mo.adaptive_weights(F, 1920, 1080, definition, hom)
    np.testing.assert_array_equal(np.asarray(lam, dtype=np.float64), g[key + '_lam'].astype(np.float64))
    # dense restatement: bit-exact (same NumPy statements as mfs.py:745-781)
    off, on = mo.jacobi_method_input(F, 1920, 1080, definition, hom, omega)
    np.testing.assert_array_equal(on, g[key + '_on'])
    band = g[key + '_band']
    for d in range(-omega, omega + 1):
        idx = np.arange(max(0, -d), min(F, F - d))
        np.testing.assert_array_equal(off[idx, idx + d], band[idx, d + omega])
    # band restatement: same numbers up to summation order
    taps, lam2, on2 = mo.jacobi_band_coefficients(F, 1920, 1080, definition, hom, omega)
    np.testing.assert_allclose(on2, g[key + '_on'], rtol=1e-14)
    ref_band = -2 * (lam2[:, None] * taps[None, :])
    valid = band != 0
    np.testing.assert_allclose(ref_band[valid], band[valid], rtol=1e-15)
    # the diagonal of `off` is NOT zero (the comment at mfs.py:749 is wrong): off[t,t] = -2*lam_t
    np.testing.assert_allclose(band[:, omega], -2 * np.asarray(lam2), rtol=1e-15)


@pytest.mark.parametrize('definition', [0, 1, 2, 3])
def test_jacobi_small_matches_reference(golden_dir, definition):
    g = _load(golden_dir, 'jacobi_small.npz')
    args = (int(g['width']), int(g['height']), definition, g['disp'], g['hom'], int(g['omega']), int(g['iters']))
    dense = mo.stabilized_vertex_displacements(*args, dense=True)
    np.testing.assert_array_equal(dense, g[f'stab_D{definition}'])          # same statements -> bit-exact
    banded = mo.stabilized_vertex_displacements(*args, dense=False)
    scale = max(1.0, np.abs(g[f'stab_D{definition}']).max())
    assert np.abs(banded - g[f'stab_D{definition}']).max() <= 1e-11 * scale


@pytest.mark.parametrize('name', ['jacobi_cfg2_subset', 'jacobi_cfg2_high_subset', 'jacobi_cfg3_subset'])
def test_jacobi_config_sized_subsets(golden_dir, name):
    g = _load(golden_dir, name + '.npz')
    F, omega, iters = int(g['F']), int(g['omega']), int(g['iters'])
    taps, lam, on = mo.jacobi_band_coefficients(F, int(g['width']), int(g['height']), int(g['definition']),
                                                g['hom'], omega)
    b = g['inputs'].reshape(F, -1)
    x = mo.jacobi_banded(b, taps, lam, on, omega, iters).reshape(g['outputs'].shape)
    scale = max(1.0, np.abs(g['outputs']).max())
    assert np.abs(x - g['outputs']).max() <= 1e-10 * scale


def test_config_inputs_are_reproducible(golden_dir):
    """The generator that produced the cfg2 inputs gives the same numbers today."""
    from meshflow_amd import synthetic
    g = _load(golden_dir, 'jacobi_cfg2_subset.npz')
    disp, hom = synthetic.motion(int(g['F']), int(g['R']), int(g['C']), seed=int(g['seed']))
    flat = disp.reshape(int(g['F']), -1, 2)[:, g['verts']]
    np.testing.assert_allclose(flat, g['inputs'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(hom, g['hom'], rtol=0, atol=1e-12)


def test_vertex_grid_matches_reference(golden_dir):
    g = _load(golden_dir, 'vertex_xy.npz')
    for key in g.files:
        W, H, R, C = (int(p[1:]) for p in key.split('_'))
        got = mo.vertex_x_y(W, H, R, C)
        assert got.dtype == np.float32 and got.shape == ((R + 1) * (C + 1), 1, 2)
        np.testing.assert_array_equal(got, g[key])


def test_stability_score_matches_reference(golden_dir):
    g = _load(golden_dir, 'stability.npz')
    for i in range(3):
        assert mo.stability_score(g[f'disp{i}']) == float(g[f'score{i}'])

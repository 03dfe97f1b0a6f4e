"""-m gpu: the HIP warp against (a) the REFERENCE's own outputs (tests/golden/warp_*.npz, written by oracle/gen_golden.py
from the real `_get_stabilized_frames_and_crop_boundaries`, mfs.py:909-1108, under a stub cv2) and (b) the C oracle on
BASELINE config 3's warp workload -- 1920x1080, 32x32 mesh, paths smoothed with omega = 30 / 200 sweeps -- where only a
third of the footprints have a single owner and the general ownership loop is busiest.  All bit-exact."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

CASES = ['warp_small', 'warp_ragged', 'warp_jitter', 'warp_shift', 'warp_mesh16']


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    return g, int(g['R']), int(g['C']), tuple(int(v) for v in g['border'])


@pytest.mark.parametrize('name', CASES)
def test_c_abi_host_wrapper_equals_the_reference(dev, golden_dir, name):
    """mf_warp_u8c3_host, raw ctypes (what INTEGRATION.md's stub calls), against the reference's frames and bounds."""
    from meshflow_amd import _lib
    g, R, C, border = _golden(golden_dir, name)
    frames = np.ascontiguousarray(g['frames'])
    n, H, W = frames.shape[:3]
    out = np.empty_like(frames)
    crop = np.zeros((n, 4), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(_lib.lib.mf_warp_u8c3_host(p(frames), p(out), p(np.ascontiguousarray(g['unstab'])),
                                          p(np.ascontiguousarray(g['stab'])), n, W, H, R, C,
                                          (ctypes.c_uint8 * 3)(*border), p(crop), None))
    np.testing.assert_array_equal(out, g['out'])
    bounds = (crop[:, 0].max(), crop[:, 1].max(), crop[:, 2].min(), crop[:, 3].min())          # mfs.py:1103-1106
    assert tuple(int(v) for v in bounds) == tuple(int(v) for v in g['bounds'])


@pytest.mark.parametrize('name', CASES)
def test_drop_in_method_equals_the_reference(dev, golden_dir, name):
    """`_get_stabilized_frames_and_crop_boundaries` with the reference's signature: list of frames in, (list, tuple) out."""
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    g, R, C, border = _golden(golden_dir, name)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, color_outside_image_area_bgr=border)
    frames, bounds = s._get_stabilized_frames_and_crop_boundaries(len(g['frames']), list(g['frames']), g['unstab'], g['stab'])
    assert isinstance(frames, list) and all(f.dtype == np.uint8 for f in frames)
    np.testing.assert_array_equal(np.stack(frames), g['out'])
    assert all(isinstance(b, np.int64) for b in bounds)
    assert tuple(int(v) for v in bounds) == tuple(int(v) for v in g['bounds'])


# ---------------------------------------------------------------------------------------------- config 3

CFG3 = dict(F=600, H=1080, W=1920, R=32, C=32, omega=30, iters=200)


@pytest.fixture(scope='module')
def cfg3(dev):
    """Config-3 motion and its smoothed paths (HIP Jacobi)."""
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    c = CFG3
    disp, hom = synthetic.motion(c['F'], c['R'], c['C'], seed=0)
    s = MeshFlowStabilizer(mesh_row_count=c['R'], mesh_col_count=c['C'], temporal_smoothing_radius=c['omega'],
                           optimization_num_iterations=c['iters'], device=str(dev))
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, c['W'], c['H'], 0, hom)
    return s, disp, hom, d_disp, d_stab


@pytest.mark.parametrize('kind', ['noise', 'pattern'])
def test_cfg3_warp_frames_vs_c_oracle(dev, cfg3, kind):
    """Four frames of the config-3 clip (first, two inner, last), records + pixels + crop values bit-identical."""
    from meshflow_amd import ops, synthetic
    from oracle import clib
    c = CFG3
    s, disp, hom, d_disp, d_stab = cfg3
    stab = d_stab.cpu().numpy()
    # the smoothed paths themselves against the C oracle's banded Jacobi (the product's O(F) coefficient set-up differs from
    # the oracle's by float64 rounding, so this is a tolerance; the kernel alone is bit-identical: test_gpu_parity.py)
    from oracle import meshflow_oracle as mo
    taps, lam, on = mo.jacobi_band_coefficients(c['F'], c['W'], c['H'], 0, hom, c['omega'])
    want_stab = clib.jacobi_banded(disp.reshape(c['F'], -1), taps, lam, np.reciprocal(on), c['omega'], c['iters'], openmp=True)
    assert np.abs(stab.reshape(c['F'], -1) - want_stab).max() <= 1e-9 * max(1.0, np.abs(want_stab).max())
    sel = [0, 201, 418, 599]
    frames = np.concatenate([synthetic.frames_numpy(1, c['H'], c['W'], seed=0, kind=kind, first_frame=f) for f in sel])
    d_fr = torch.from_numpy(frames).to(dev)
    table = ops.cell_table(d_disp[sel], d_stab[sel], c['W'], c['H'], c['R'], c['C'])
    out = ops.warp(d_fr, table, (0, 0, 255))
    torch.cuda.synchronize()
    table.check()
    want, want_crop, bad = clib.warp_clip(frames, c['R'], c['C'], disp[sel], stab[sel], use_bbox=True, openmp=True)
    assert bad == 0
    rec = table.records().cpu().numpy()
    for i, f in enumerate(sel):
        tab, _ = clib.cell_table(c['W'], c['H'], c['R'], c['C'], disp[f], stab[f])
        np.testing.assert_array_equal(rec[i], tab)
    got = out.cpu().numpy()
    assert np.array_equal(got, want), f'{(got != want).sum()} bytes differ'
    np.testing.assert_array_equal(table.crop.cpu().numpy(), want_crop)


def test_cfg3_full_clip_properties(dev, cfg3):
    """All 600 frames of 1920x1080 with the 32x32 mesh, device-resident: integer global shift (every interior pixel moves
    by exactly the shift, uncovered band in the border colour, analytic crop bounds), determinism of the smoothed-motion
    warp over two launches, a sample of frames against the C oracle, clip-level bounds = mfs.py:1103-1106."""
    from meshflow_amd import ops, synthetic
    from oracle import clib
    c = CFG3
    s, disp, hom, d_disp, d_stab = cfg3
    F, H, W, R, C = c['F'], c['H'], c['W'], c['R'], c['C']
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
    dx, dy = 11, 6
    d_shift = d_disp + torch.tensor([dx, dy], dtype=torch.float64, device=dev)
    table = ops.cell_table(d_disp, d_shift, W, H, R, C)
    out = ops.warp(d_frames, table, (9, 8, 7))
    table.check()
    assert torch.equal(out[:, dy:, dx:], d_frames[:, :H - dy, :W - dx])
    border = torch.tensor([9, 8, 7], dtype=torch.uint8, device=dev)
    assert bool((out[:, :, :dx - 1] == border).all()) and bool((out[:, :dy - 1] == border).all())
    assert ops.crop_reduce(table.crop, W, H).tolist() == [dx, dy, W - 1, H - 1]
    del out, table
    out1, crop1 = s._stabilized_frames_device(d_frames, d_disp, d_stab)
    out1, crop1 = out1.clone(), crop1.clone()
    out2, crop2 = s._stabilized_frames_device(d_frames, d_disp, d_stab)
    assert torch.equal(out1, out2) and torch.equal(crop1, crop2)
    del out2
    sel = [7, 300, 592]
    stab = d_stab.cpu().numpy()
    want, want_crop, bad = clib.warp_clip(d_frames[sel].cpu().numpy(), R, C, disp[sel], stab[sel], use_bbox=True, openmp=True)
    assert bad == 0
    assert np.array_equal(out1[sel].cpu().numpy(), want)
    np.testing.assert_array_equal(crop1[sel].cpu().numpy(), want_crop)
    crop_h = crop1.cpu().numpy()
    assert ops.crop_reduce(crop1, W, H).tolist() == [crop_h[:, 0].max(), crop_h[:, 1].max(), crop_h[:, 2].min(), crop_h[:, 3].min()]

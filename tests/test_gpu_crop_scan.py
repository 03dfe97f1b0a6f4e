"""-m gpu: the scan-only crop pass (`mf_crop_scan_f64`, csrc/warp.hip crop_scan_kernel).

The four per-frame edge scans of the reference (mfs.py:1075-1098) and the clip-level rectangle (mfs.py:1103-1106) look at the
coordinate maps only -- at nothing but the cell table -- so they can be had BEFORE a single pixel is warped.  The pass must fill
the per-frame values exactly as the warp kernel's fused scan does (which the other GPU tests hold to the oracle and to the
reference's own goldens): small and odd geometries, stress geometries, the committed goldens, and BASELINE's full sizes."""
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


def _scan_and_warp(dev, frames, R, C, unstab, stab, border=(0, 0, 255)):
    """(crop from the scan-only pass, crop after the warp ran on the SAME table (nothing may change), crop of a warp alone, out)."""
    from meshflow_amd import ops
    n, H, W = frames.shape[:3]
    d_un = torch.from_numpy(np.ascontiguousarray(unstab)).to(dev)
    d_st = torch.from_numpy(np.ascontiguousarray(stab)).to(dev)
    d_fr = torch.from_numpy(frames).to(dev)
    table = ops.cell_table(d_un, d_st, W, H, R, C)
    scanned = ops.crop_scan(table).clone()
    # the clip-level rectangle the kernels fold together inside the table blob (no reduction launch) equals mf_crop_reduce's
    assert torch.equal(table.clip_bounds, ops.crop_reduce(scanned, W, H))
    out = ops.warp(d_fr, table, border)
    after = table.crop.clone()
    assert torch.equal(table.clip_bounds, ops.crop_reduce(after, W, H))
    table2 = ops.cell_table(d_un, d_st, W, H, R, C)
    assert table2.clip_bounds.tolist() == [0, 0, W - 1, H - 1]
    ops.warp(d_fr, table2, border)
    assert torch.equal(table2.clip_bounds, ops.crop_reduce(table2.crop, W, H))
    torch.cuda.synchronize()
    table.check()
    return scanned.cpu().numpy(), after.cpu().numpy(), table2.crop.cpu().numpy(), out.cpu().numpy()


def _clip(F, H, W, R, C, seed, kind='noise', omega=3, iters=10, **kw):
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind=kind, **kw)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, omega, iters)
    return frames, disp, stab


@pytest.mark.parametrize('H,W,R,C,kw', [
    (64, 96, 4, 4, dict(jitter_sigma=1.0)),
    (150, 200, 3, 5, dict(jitter_sigma=0.5)),
    (75, 101, 3, 5, dict(jitter_sigma=0.5)),                        # W % 4 != 0: nothing is staged, every footprint is scanned
    (130, 260, 16, 16, dict(jitter_sigma=0.3)),
    (96, 160, 32, 32, dict(translation_sigma=1.0, field_sigma=0.3)),
    (360, 640, 16, 16, dict(translation_sigma=8.0, jitter_sigma=2.0)),
    (272, 480, 3, 4, dict(jitter_sigma=0.5)),
    (272, 480, 2, 2, dict(translation_sigma=3.0, jitter_sigma=1.5)),
    (17, 23, 2, 3, dict(jitter_sigma=0.5)),                         # frame smaller than a footprint
    (2400, 64, 12, 2, dict(jitter_sigma=0.5)),                      # two footprints per row, 300 rows: more footprint rows in a plan tile than its row table holds
    (1200, 128, 6, 3, dict(translation_sigma=2.0, jitter_sigma=0.8)),   # 150 rows of four: a plan tile (1024 footprints) ends in the middle of the frame
])
def test_scan_equals_fused_scan_and_oracle(dev, H, W, R, C, kw):
    from oracle import clib
    frames, disp, stab = _clip(6, H, W, R, C, seed=H + 3 * W, **kw)
    scanned, after, fused, out = _scan_and_warp(dev, frames, R, C, disp, stab)
    want, want_crop, bad = clib.warp_clip(frames, R, C, disp, stab)
    assert bad == 0
    np.testing.assert_array_equal(fused, want_crop)
    np.testing.assert_array_equal(scanned, want_crop)
    np.testing.assert_array_equal(after, want_crop)                 # the warp's own scan on top of it changes nothing
    np.testing.assert_array_equal(out, want)


@pytest.mark.parametrize('H,W,R,C,sigma,seed', [
    (130, 260, 4, 4, 8.0, 1),       # strongly non-affine quads
    (130, 260, 4, 4, 20.0, 2),      # folded quads
    (64, 96, 8, 8, 6.0, 3),         # more than 8 candidate cells per footprint: range-scan path
    (200, 300, 64, 64, 0.4, 4),
])
def test_scan_stress_geometries(dev, H, W, R, C, sigma, seed):
    from meshflow_amd import synthetic
    from oracle import clib
    frames = synthetic.frames_numpy(2, H, W, seed=seed, kind='noise')
    n = np.arange(2 * (R + 1) * (C + 1) * 2, dtype=np.int64).reshape(2, R + 1, C + 1, 2)
    unstab = np.zeros((2, R + 1, C + 1, 2))
    stab = sigma * synthetic.normal(n, seed=100 + seed)
    want, want_crop, bad = clib.warp_clip(frames, R, C, unstab, stab)
    if bad:
        pytest.skip('degenerate mesh (covered by test_gpu_parity.py)')
    scanned, after, fused, _ = _scan_and_warp(dev, frames, R, C, unstab, stab)
    np.testing.assert_array_equal(scanned, want_crop)
    np.testing.assert_array_equal(after, want_crop)
    np.testing.assert_array_equal(fused, want_crop)


@pytest.mark.parametrize('name', ['warp_small', 'warp_ragged', 'warp_jitter', 'warp_shift', 'warp_mesh16'])
def test_scan_gives_the_reference_rectangle(dev, golden_dir, name):
    """Raw C ABI on the reference's own goldens (oracle/gen_golden.py ran mfs.py:909-1108): cell table -> scan -> reduce gives the
    reference's crop boundaries without any frame on the device."""
    from meshflow_amd import _lib
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    R, C = int(g['R']), int(g['C'])
    n, H, W = g['frames'].shape[:3]
    lib = _lib.lib
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_un, d_st = t(g['unstab']), t(g['stab'])
    table = torch.empty(lib.mf_cell_table_bytes(n, W, H, R, C), dtype=torch.uint8, device=dev)
    crop = torch.empty((n, 4), dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    bounds = torch.empty(4, dtype=torch.int32, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.mf_cell_table_f64(p(d_un), p(d_st), n, W, H, R, C, p(table), p(crop), p(status), st))
    _lib.check(lib.mf_crop_scan_f64(p(table), n, W, H, R, C, p(crop), st))
    _lib.check(lib.mf_crop_reduce(p(crop), n, W, H, p(bounds), st))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert tuple(bounds.tolist()) == tuple(int(v) for v in g['bounds'])
    if 'per_frame' in g.files:
        np.testing.assert_array_equal(crop.cpu().numpy(), g['per_frame'])


@pytest.mark.parametrize('F,H,W,R,C,omega,iters,clip_frames,first', [
    (300, 1080, 1920, 16, 16, 10, 100, 300, 0),          # BASELINE config 2
    (600, 1080, 1920, 32, 32, 30, 200, 600, 0),          # config 3
    (150, 2160, 3840, 16, 16, 10, 100, 1200, 450),       # config 4 at shard size
])
def test_scan_full_size_clips(dev, F, H, W, R, C, omega, iters, clip_frames, first):
    """BASELINE's sizes, device-resident: the scan of the whole clip equals the warp kernel's fused scan frame for frame (the
    warp's values are held to the C oracle on sampled frames elsewhere), an integer global shift gives the analytic rectangle
    without a pixel, and only a few per cent of the footprints are visited."""
    from meshflow_amd import ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    disp_all, hom = synthetic.motion(clip_frames, R, C, seed=0)
    d_disp_all = torch.from_numpy(disp_all).to(dev)
    d_disp = d_disp_all[first:first + F]
    dx, dy = 7, -5
    table = ops.cell_table(d_disp, d_disp + torch.tensor([dx, dy], dtype=torch.float64, device=dev), W, H, R, C)
    assert ops.crop_reduce(ops.crop_scan(table), W, H).tolist() == [dx, 0, W - 1, H - 1 + dy]
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=omega, optimization_num_iterations=iters,
                           device=str(dev))
    d_stab = s._stabilized_vertex_displacements_device(d_disp_all, W, H, 0, hom)[first:first + F]
    table = ops.cell_table(d_disp, d_stab, W, H, R, C, table=table)
    scanned = ops.crop_scan(table).clone()
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern', first_frame=first)
    table2 = ops.cell_table(d_disp, d_stab, W, H, R, C)
    ops.warp(d_frames, table2, (0, 0, 255))
    torch.cuda.synchronize()
    table2.check()
    assert torch.equal(scanned, table2.crop)
    assert (scanned != torch.tensor([0, 0, W - 1, H - 1], dtype=torch.int32, device=dev)).any()      # the clip does crop
    # share of the footprints the scan has to visit (region words sit behind the plan in the table blob; layout: csrc/mf_common.h)
    nfp = F * (-(-H // 8)) * (-(-W // 32))
    nrec = F * R * C
    plan_off = (nrec * (32 * 8 + 8 + (16 + 12) * 4) + 15) & ~15
    regions = table.buf[plan_off + 16 * nfp: plan_off + 24 * nfp].view(torch.int32).view(nfp, 2)[:, 0]
    visited = int(((regions & 0x50000000) == 0).sum().item())
    assert visited / nfp < 0.12, visited / nfp


def test_scan_argument_checks(dev):
    from meshflow_amd import _lib
    assert _lib.lib.mf_crop_scan_f64(None, 1, 64, 64, 4, 4, None, None) == _lib.MF_ERR_INVALID_ARG
    assert b'null' in _lib.lib.mf_last_error()

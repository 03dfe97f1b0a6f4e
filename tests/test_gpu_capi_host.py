"""-m gpu: the host-buffer and multi-GPU entry points of the C ABI, raw ctypes (no torch in the calls).

mf_warp_u8c3_host / mf_warp_u8c3_host_frames: chunked, overlapped PCIe staging below Python (csrc/hostpipe.hip) -- several
chunks, ragged last chunk, cache reuse across calls and shapes, separate per-frame allocations, pinned buffers, errors.
mf_comm_init_all / mf_allreduce_crop / mf_gather_frames / mf_comm_destroy: RCCL directly (csrc/comm.hip); on the one-GPU
test box the communicator has one rank (the collectives then run device-to-device on that GPU)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: the -m gpu tests must run on an MI355X')
    return torch.device('cuda:0')


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _clip(F, H, W, R, C, seed, **kw):
    from meshflow_amd import synthetic
    from oracle import meshflow_oracle as mo
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=seed, kind='noise', **kw)
    stab = mo.stabilized_vertex_displacements(W, H, 0, disp, hom, 3, 10)
    return frames, np.ascontiguousarray(disp), np.ascontiguousarray(stab)


@pytest.mark.parametrize('F,H,W,R,C', [(40, 72, 100, 3, 5), (16, 64, 96, 4, 4), (50, 96, 128, 8, 8), (3, 48, 64, 2, 2)])
def test_host_wrapper_chunks_equal_the_oracle(F, H, W, R, C):
    from meshflow_amd import _lib
    from oracle import clib
    frames, disp, stab = _clip(F, H, W, R, C, seed=F + W, jitter_sigma=0.8)
    want, want_crop, bad = clib.warp_clip(frames, R, C, disp, stab, (5, 6, 7))
    assert bad == 0
    border = (ctypes.c_uint8 * 3)(5, 6, 7)
    for _ in range(2):                                      # second call: cached device buffers and streams
        out = np.zeros_like(frames)
        crop = np.zeros((F, 4), np.int32)
        ms = ctypes.c_float(-1)
        _lib.check(_lib.lib.mf_warp_u8c3_host(_p(frames), _p(out), _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), ctypes.byref(ms)))
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(crop, want_crop)
        assert ms.value > 0
    # separate per-frame allocations, as the reference's frame lists are (mfs.py:997, 1100)
    ins = [f.copy() for f in frames]
    outs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
    pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in ins])
    pout = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs])
    crop = np.zeros((F, 4), np.int32)
    _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), None))
    np.testing.assert_array_equal(np.stack(outs), want)
    np.testing.assert_array_equal(crop, want_crop)


def test_host_wrapper_pinned_buffers_and_cache_release():
    from meshflow_amd import _lib
    from oracle import clib
    F, H, W, R, C = 37, 64, 96, 4, 4
    frames, disp, stab = _clip(F, H, W, R, C, seed=3)
    nbytes = frames.nbytes
    hin, hout = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(_lib.lib.mf_malloc_host(ctypes.byref(hin), nbytes))
    _lib.check(_lib.lib.mf_malloc_host(ctypes.byref(hout), nbytes))
    try:
        ctypes.memmove(hin, frames.ctypes.data, nbytes)
        crop = np.zeros((F, 4), np.int32)
        _lib.check(_lib.lib.mf_warp_u8c3_host(hin, hout, _p(disp), _p(stab), F, W, H, R, C, (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), None))
        out = np.ctypeslib.as_array(ctypes.cast(hout, ctypes.POINTER(ctypes.c_uint8)), shape=frames.shape).copy()
    finally:
        _lib.check(_lib.lib.mf_free_host(hin))
        _lib.check(_lib.lib.mf_free_host(hout))
    want, want_crop, _ = clib.warp_clip(frames, R, C, disp, stab)
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(crop, want_crop)
    assert _lib.lib.mf_host_cache_release() == 0 and _lib.lib.mf_host_cache_release() == 0
    out2 = np.zeros_like(frames)                            # works again after the release
    _lib.check(_lib.lib.mf_warp_u8c3_host(_p(frames), _p(out2), _p(disp), _p(stab), F, W, H, R, C, (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), None))
    np.testing.assert_array_equal(out2, want)


def test_host_wrapper_reports_degenerate_mesh_and_recovers():
    from meshflow_amd import _lib
    F, H, W, R, C = 20, 64, 96, 4, 4
    frames, disp, stab = _clip(F, H, W, R, C, seed=9)
    bad = stab.copy()
    grid_x = np.array([np.ceil((W - 1) * c / C) for c in range(C + 1)])
    bad[17, :, :, 0] = disp[17, :, :, 0] - grid_x[None, :]          # frame 17 (second chunk): every vertex onto x = 0
    out = np.zeros_like(frames)
    crop = np.zeros((F, 4), np.int32)
    rc = _lib.lib.mf_warp_u8c3_host(_p(frames), _p(out), _p(disp), _p(bad), F, W, H, R, C, (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), None)
    assert rc == _lib.MF_ERR_DEGENERATE and b'degenerate' in _lib.lib.mf_last_error()
    assert _lib.lib.mf_warp_u8c3_host(None, _p(out), _p(disp), _p(stab), F, W, H, R, C, (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), None) == _lib.MF_ERR_INVALID_ARG
    _lib.check(_lib.lib.mf_warp_u8c3_host(_p(frames), _p(out), _p(disp), _p(stab), F, W, H, R, C, (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), None))


def test_rccl_entry_points_single_rank():
    """ncclCommInitAll over the visible GPU(s); the crop all-reduce and the ragged frame gather through the C ABI."""
    from meshflow_amd import _lib
    lib = _lib.lib
    count = ctypes.c_int(0)
    _lib.check(lib.mf_device_count(ctypes.byref(count)))
    n = ctypes.c_int(-1)
    _lib.check(lib.mf_comm_size(ctypes.byref(n)))
    assert n.value == 0
    assert lib.mf_allreduce_crop((ctypes.c_void_p * 1)(None)) == _lib.MF_ERR_INVALID_ARG      # no communicator yet
    assert lib.mf_comm_init_all(count.value + 1) == _lib.MF_ERR_INVALID_ARG
    ndev = count.value
    _lib.check(lib.mf_comm_init_all(ndev))
    try:
        assert lib.mf_comm_init_all(ndev) == _lib.MF_ERR_INVALID_ARG                          # already initialised
        _lib.check(lib.mf_comm_size(ctypes.byref(n)))
        assert n.value == ndev
        bounds, shards, sizes = [], [], []
        rng = np.random.default_rng(1)
        host_bounds = rng.integers(0, 1000, size=(ndev, 4)).astype(np.int32)
        host_shards = [rng.integers(0, 256, size=1000 + 37 * g, dtype=np.uint8) for g in range(ndev)]
        for g in range(ndev):
            _lib.check(lib.mf_set_device(g))
            b, s = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(lib.mf_malloc(ctypes.byref(b), 16)); _lib.check(lib.mf_malloc(ctypes.byref(s), host_shards[g].nbytes))
            _lib.check(lib.mf_memcpy_h2d(b, _p(host_bounds[g]), 16, None)); _lib.check(lib.mf_memcpy_h2d(s, _p(host_shards[g]), host_shards[g].nbytes, None))
            _lib.check(lib.mf_stream_synchronize(None))
            bounds.append(b); shards.append(s); sizes.append(host_shards[g].nbytes)
        _lib.check(lib.mf_allreduce_crop((ctypes.c_void_p * ndev)(*[b.value for b in bounds])))
        want = np.concatenate([host_bounds[:, :2].max(0), host_bounds[:, 2:].min(0)])
        for g in range(ndev):
            _lib.check(lib.mf_set_device(g))
            got = np.zeros(4, np.int32)
            _lib.check(lib.mf_memcpy_d2h(_p(got), bounds[g], 16, None)); _lib.check(lib.mf_stream_synchronize(None))
            np.testing.assert_array_equal(got, want)
        _lib.check(lib.mf_set_device(0))
        total = sum(sizes)
        dst = ctypes.c_void_p()
        _lib.check(lib.mf_malloc(ctypes.byref(dst), total))
        _lib.check(lib.mf_gather_frames((ctypes.c_void_p * ndev)(*[s.value for s in shards]), (ctypes.c_size_t * ndev)(*sizes), dst, 0))
        got = np.zeros(total, np.uint8)
        _lib.check(lib.mf_memcpy_d2h(_p(got), dst, total, None)); _lib.check(lib.mf_stream_synchronize(None))
        np.testing.assert_array_equal(got, np.concatenate(host_shards))
        _lib.check(lib.mf_free(dst))
        for g in range(ndev):
            _lib.check(lib.mf_set_device(g)); _lib.check(lib.mf_free(bounds[g])); _lib.check(lib.mf_free(shards[g]))
        _lib.check(lib.mf_set_device(0))
    finally:
        _lib.check(lib.mf_comm_destroy())
    _lib.check(lib.mf_comm_size(ctypes.byref(n)))
    assert n.value == 0


@pytest.mark.parametrize('F,H,W,R,C,keep', [(40, 72, 100, 3, 5, True), (21, 96, 128, 8, 8, False)])
def test_host_warp_crop_pipeline_equals_oracle(F, H, W, R, C, keep):
    """mf_warp_crop_u8c3_host_frames = warp (mfs.py:909-1108) + clip-level rectangle (mfs.py:1103-1106) + _crop_frames
    (mfs.py:1111-1157) in ONE host-to-host pipeline: cropped frames equal the NumPy oracle's crop of the C oracle's warp."""
    from meshflow_amd import _lib
    from oracle import clib, meshflow_oracle as mo
    frames, disp, stab = _clip(F, H, W, R, C, seed=F + H)
    want, want_crop, bad = clib.warp_clip(frames, R, C, disp, stab, (9, 8, 7))
    assert bad == 0
    rect = (want_crop[:, 0].max(), want_crop[:, 1].max(), want_crop[:, 2].min(), want_crop[:, 3].min())
    want_cropped = np.stack(mo.crop_frames(list(want), rect))
    border = (ctypes.c_uint8 * 3)(9, 8, 7)
    ins = [f.copy() for f in frames]
    outs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
    crs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
    pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in ins])
    pout = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs]) if keep else None
    pcr = (ctypes.c_void_p * F)(*[f.ctypes.data for f in crs])
    for _ in range(2):                                      # second call: cached buffers
        crop = np.zeros((F, 4), np.int32)
        bounds = (ctypes.c_int32 * 4)()
        ms = ctypes.c_float(-1)
        _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, pout, pcr, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), bounds,
                                                          ctypes.byref(ms)))
        assert tuple(bounds) == tuple(int(v) for v in rect)
        np.testing.assert_array_equal(crop, want_crop)
        np.testing.assert_array_equal(np.stack(crs), want_cropped)
        if keep:
            np.testing.assert_array_equal(np.stack(outs), want)
        assert ms.value > 0
    np.testing.assert_array_equal(np.stack(ins), frames)    # inputs untouched


def test_host_wrapper_rejects_overlapping_input_and_output():
    from meshflow_amd import _lib
    F, H, W, R, C = 8, 48, 64, 2, 2
    frames, disp, stab = _clip(F, H, W, R, C, seed=2)
    crop = np.zeros((F, 4), np.int32)
    border = (ctypes.c_uint8 * 3)(0, 0, 255)
    rc = _lib.lib.mf_warp_u8c3_host(_p(frames), _p(frames), _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), None)
    assert rc == _lib.MF_ERR_INVALID_ARG and b'overlap' in _lib.lib.mf_last_error()
    big = np.zeros((F + 1, H, W, 3), np.uint8)                # output shifted by one frame over the input
    big[:F] = frames
    pin = (ctypes.c_void_p * F)(*[big[i].ctypes.data for i in range(F)])
    pout = (ctypes.c_void_p * F)(*[big[i + 1].ctypes.data for i in range(F)])
    rc = _lib.lib.mf_warp_u8c3_host_frames(pin, pout, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), None)
    assert rc == _lib.MF_ERR_INVALID_ARG


def test_host_warp_crop_reports_empty_rectangle():
    """A clip whose crop rectangle is empty (cv2.resize would fail on an empty source): MF_ERR_INVALID_ARG, bounds still written."""
    from meshflow_amd import _lib, synthetic
    F, H, W, R, C = 6, 64, 96, 4, 4
    frames = synthetic.frames_numpy(F, H, W, seed=1)
    disp = np.zeros((F, R + 1, C + 1, 2))
    stab = disp.copy()
    stab[0, ..., 0] = 60.0       # content moves right by 60 px in frame 0 ...
    stab[1, ..., 0] = -60.0      # ... and left by 60 px in frame 1: left bound 60 > right bound W - 1 - 60
    border = (ctypes.c_uint8 * 3)(0, 0, 255)
    crs = np.full_like(frames, 0xAB)
    fb = H * W * 3
    pin = (ctypes.c_void_p * F)(*[frames.ctypes.data + i * fb for i in range(F)])
    pcr = (ctypes.c_void_p * F)(*[crs.ctypes.data + i * fb for i in range(F)])
    crop = np.zeros((F, 4), np.int32)
    bounds = (ctypes.c_int32 * 4)()
    rc = _lib.lib.mf_warp_crop_u8c3_host_frames(pin, None, pcr, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), bounds, None)
    assert rc == _lib.MF_ERR_INVALID_ARG and b'empty crop rectangle' in _lib.lib.mf_last_error()
    assert bounds[0] > bounds[2]
    # the rectangle comes from the cell table alone (mf_crop_scan_f64), before any frame moves: no output byte has been touched, and
    # the per-frame values say which frames emptied it
    assert (crs == 0xAB).all()
    assert crop[0].tolist() == [60, 0, W - 1, H - 1] and crop[1].tolist() == [0, 0, W - 1 - 60, H - 1]
    # and the pipeline works again afterwards
    stab[:] = 0.0
    _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, None, pcr, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), bounds, None))
    np.testing.assert_array_equal(crs, frames)              # identity warp, full-frame rectangle: resize is the identity


def test_full_cfg2_clip_through_raw_c_abi_equals_device_path(dev):
    """BASELINE config 2 at full size (300 frames of 1920x1080, 16x16 mesh) through raw-ctypes mf_warp_u8c3_host_frames equals the
    device-resident operators (mf_cell_table_f64 + mf_warp_u8c3) byte for byte, and sampled frames equal the C oracle."""
    import torch
    from meshflow_amd import _lib, ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    from oracle import clib
    H, W, F, R, C = 1080, 1920, 300, 16, 16
    disp, hom = synthetic.motion(F, R, C, seed=0)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100)
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    stab = d_stab.cpu().numpy()
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
    frames = d_frames.cpu().numpy()
    table = ops.cell_table(d_disp, d_stab, W, H, R, C)
    d_out = ops.warp(d_frames, table)
    want, want_crop = d_out.cpu().numpy(), table.crop.cpu().numpy()
    del d_frames, d_out, table
    out = np.empty_like(frames)
    fb = H * W * 3
    pin = (ctypes.c_void_p * F)(*[frames.ctypes.data + i * fb for i in range(F)])
    pout = (ctypes.c_void_p * F)(*[out.ctypes.data + i * fb for i in range(F)])
    crop = np.zeros((F, 4), np.int32)
    _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout, _p(np.ascontiguousarray(disp)), _p(np.ascontiguousarray(stab)), F, W, H, R, C,
                                                 (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), None))
    np.testing.assert_array_equal(crop, want_crop)
    assert np.array_equal(out, want)
    sel = [0, 149, 299]
    ref, ref_crop, bad = clib.warp_clip(frames[sel], R, C, disp[sel], stab[sel], use_bbox=True, openmp=True)
    assert bad == 0
    np.testing.assert_array_equal(crop[sel], ref_crop)
    assert np.array_equal(out[sel], ref)


def test_full_cfg2_clip_crop_pipeline_equals_device_path(dev):
    """BASELINE config 2 at full size through mf_warp_crop_u8c3_host_frames -- rectangle from the scan-only pass, then every chunk
    uploaded, warped, cropped + resized and downloaded in ONE phase -- equals the device-resident operators (warp of the whole clip,
    rectangle from the warp's own scan, one crop + resize of the whole clip) byte for byte; the uncropped frames never come back."""
    import torch
    from meshflow_amd import _lib, ops, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    H, W, F, R, C = 1080, 1920, 300, 16, 16
    disp, hom = synthetic.motion(F, R, C, seed=0)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=10, optimization_num_iterations=100)
    d_disp = torch.from_numpy(disp).to(dev)
    d_stab = s._stabilized_vertex_displacements_device(d_disp, W, H, 0, hom)
    stab = d_stab.cpu().numpy()
    d_frames = synthetic.frames_torch(F, H, W, dev, seed=0, kind='pattern')
    frames = d_frames.cpu().numpy()
    table = ops.cell_table(d_disp, d_stab, W, H, R, C)
    d_out = ops.warp(d_frames, table)
    rect = ops.crop_reduce(table.crop, W, H).tolist()
    assert rect[0] > 0 and rect[2] < W - 1                   # the clip really crops
    want = ops.crop_resize(d_out, rect).cpu().numpy()
    want_crop = table.crop.cpu().numpy()
    del d_frames, d_out, table
    torch.cuda.empty_cache()
    got = np.empty_like(frames)
    fb = H * W * 3
    pin = (ctypes.c_void_p * F)(*[frames.ctypes.data + i * fb for i in range(F)])
    pcr = (ctypes.c_void_p * F)(*[got.ctypes.data + i * fb for i in range(F)])
    crop = np.zeros((F, 4), np.int32)
    bounds = (ctypes.c_int32 * 4)()
    for _ in range(2):
        _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, None, pcr, _p(np.ascontiguousarray(disp)), _p(np.ascontiguousarray(stab)), F, W, H, R, C,
                                                          (ctypes.c_uint8 * 3)(0, 0, 255), _p(crop), bounds, None))
        assert list(bounds) == rect
        np.testing.assert_array_equal(crop, want_crop)
        assert np.array_equal(got, want)
        got[:] = 0


def test_host_pipeline_leaves_the_current_device_alone(dev):
    """The C pipeline works on the calling thread's current HIP device; the Python boundary scopes it (torch.cuda.device) and must
    leave torch's current device as it found it -- also for device='cuda' (no index = the CURRENT device, not device 0)."""
    import torch
    from meshflow_amd import synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    F, H, W, R, C = 6, 48, 64, 2, 2
    frames, disp, hom = synthetic.clip(F, H, W, R, C, seed=5)
    last = torch.cuda.device_count() - 1
    for name in ('cuda', f'cuda:{last}'):
        s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, temporal_smoothing_radius=3, optimization_num_iterations=5, device=name)
        before = torch.cuda.current_device()
        out, bounds, stab, score, cropped = s.stabilize_clip(list(frames), disp, hom, crop=True)
        assert torch.cuda.current_device() == before
        assert s._torch_device().index == (before if name == 'cuda' else last)
        assert len(out) == F and len(cropped) == F

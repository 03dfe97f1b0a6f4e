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


# (the last two: frame sizes that are no multiple of 4 bytes -- ring slots then start at unaligned device addresses, the warp takes its
# unstaged path -- and a single-row mesh)
@pytest.mark.parametrize('F,H,W,R,C', [(40, 72, 100, 3, 5), (16, 64, 96, 4, 4), (50, 96, 128, 8, 8), (3, 48, 64, 2, 2), (11, 50, 101, 3, 4), (7, 33, 67, 1, 6), (1, 40, 64, 2, 2), (2, 9, 5, 1, 1)])
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


@pytest.mark.parametrize('F,H,W,R,C,keep', [(40, 72, 100, 3, 5, True), (21, 96, 128, 8, 8, False), (1, 40, 64, 2, 2, True), (2, 17, 9, 1, 2, False)])
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


def test_host_pipeline_any_thread_chunk_and_ring_setting_gives_the_same_bytes():
    """The ring hands the next chunk to whichever copy thread is free: twelve random settings of upload / download / populate threads,
    frames per chunk and ring slots (more threads than chunks, one slot more than two, one chunk for everything ...) all give the
    oracle's frames, cropped frames, per-frame values and rectangle."""
    import os
    from meshflow_amd import _lib
    from oracle import clib, meshflow_oracle as mo
    F, H, W, R, C = 37, 96, 128, 5, 6
    frames, disp, stab = _clip(F, H, W, R, C, seed=77)
    want, want_crop, bad = clib.warp_clip(frames, R, C, disp, stab, (1, 2, 3))
    assert bad == 0
    rect = (want_crop[:, 0].max(), want_crop[:, 1].max(), want_crop[:, 2].min(), want_crop[:, 3].min())
    want_cropped = np.stack(mo.crop_frames(list(want), rect))
    border = (ctypes.c_uint8 * 3)(1, 2, 3)
    ins = [f.copy() for f in frames]
    pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in ins])
    names = ('MF_PIPE_UP', 'MF_PIPE_DOWN', 'MF_PIPE_POPULATE', 'MF_PIPE_CHUNK', 'MF_PIPE_SLOTS')
    old = {k: os.environ.get(k) for k in names}
    rng = np.random.default_rng(5)
    settings = [(1, 1, 0, 1, 2), (8, 8, 8, 37, 2), (8, 1, 1, 3, 3), (1, 8, 2, 2, 64)] + \
               [tuple(int(v) for v in (rng.integers(1, 9), rng.integers(1, 9), rng.integers(0, 9), rng.integers(1, 12), rng.integers(2, 20))) for _ in range(8)]
    try:
        for setting in settings:
            for k, v in zip(names, setting):
                os.environ[k] = str(v)
            outs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
            crs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
            pout = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs])
            pcr = (ctypes.c_void_p * F)(*[f.ctypes.data for f in crs])
            crop = np.zeros((F, 4), np.int32)
            bounds = (ctypes.c_int32 * 4)()
            _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, pout, pcr, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), bounds, None))
            assert tuple(bounds) == tuple(int(v) for v in rect), setting
            np.testing.assert_array_equal(crop, want_crop, err_msg=str(setting))
            np.testing.assert_array_equal(np.stack(crs), want_cropped, err_msg=str(setting))
            np.testing.assert_array_equal(np.stack(outs), want, err_msg=str(setting))
            outs2 = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]                         # the warp alone through the same ring
            pout2 = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs2])
            _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout2, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), None))
            np.testing.assert_array_equal(np.stack(outs2), want, err_msg=str(setting))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    np.testing.assert_array_equal(np.stack(ins), frames)


def test_host_pipeline_with_frames_that_are_no_multiple_of_four_bytes(monkeypatch):
    """101 x 50 frames (15,150 bytes) in chunks of 3 on a ring of 2 slots: every second slot starts at a device address that is not
    4-byte aligned -- the warp must take its unstaged path there and the crop + resize its direct one; results as the oracle's."""
    from meshflow_amd import _lib
    from oracle import clib, meshflow_oracle as mo
    F, H, W, R, C = 11, 50, 101, 3, 4
    frames, disp, stab = _clip(F, H, W, R, C, seed=9, jitter_sigma=0.8)
    want, want_crop, bad = clib.warp_clip(frames, R, C, disp, stab, (5, 6, 7))
    assert bad == 0
    rect = (want_crop[:, 0].max(), want_crop[:, 1].max(), want_crop[:, 2].min(), want_crop[:, 3].min())
    want_cropped = np.stack(mo.crop_frames(list(want), rect))
    monkeypatch.setenv('MF_PIPE_CHUNK', '3')
    monkeypatch.setenv('MF_PIPE_SLOTS', '2')
    border = (ctypes.c_uint8 * 3)(5, 6, 7)
    ins = [f.copy() for f in frames]
    outs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
    crs = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
    pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in ins])
    pout = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs])
    pcr = (ctypes.c_void_p * F)(*[f.ctypes.data for f in crs])
    crop = np.zeros((F, 4), np.int32)
    bounds = (ctypes.c_int32 * 4)()
    _lib.check(_lib.lib.mf_warp_crop_u8c3_host_frames(pin, pout, pcr, _p(disp), _p(stab), F, W, H, R, C, border, _p(crop), bounds, None))
    assert tuple(bounds) == tuple(int(v) for v in rect)
    np.testing.assert_array_equal(crop, want_crop)
    np.testing.assert_array_equal(np.stack(outs), want)
    np.testing.assert_array_equal(np.stack(crs), want_cropped)
    outs2 = [np.zeros((H, W, 3), np.uint8) for _ in range(F)]
    pout2 = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs2])
    _lib.check(_lib.lib.mf_crop_resize_u8c3_host_frames(pout, pout2, F, W, H, *[int(v) for v in rect], None))
    np.testing.assert_array_equal(np.stack(outs2), want_cropped)


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


# ---- the ring of chunk buffers: device memory O(chunk), clips of any length ----

def _hbm_used():
    import torch
    free_b, total_b = torch.cuda.mem_get_info(0)
    return total_b - free_b


def _device_warp_crop(d_frames, disp, stab, R, C, rect=None):
    """The device operators on resident frames (what the host pipeline must reproduce chunk by chunk): stabilized frames, per-frame
    crop rows, the clip rectangle and the cropped + resized frames."""
    import torch
    from meshflow_amd import ops
    n, H, W, _ = d_frames.shape
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d_frames.device)
    table = ops.cell_table(t(disp), t(stab), W, H, R, C)
    out = ops.warp(d_frames, table)
    crop = table.crop.cpu().numpy()
    if rect is None:
        rect = (int(crop[:, 0].max()), int(crop[:, 1].max()), int(crop[:, 2].min()), int(crop[:, 3].min()))
    return out, crop, rect, ops.crop_resize(out, rect)


def test_crop_resize_host_frames_raw_ctypes_and_errors(dev):
    """mf_crop_resize_u8c3_host_frames: _crop_frames (mfs.py:1111-1157) by itself through the ring -- several chunks, a ring of two slots,
    separate per-frame allocations; bad rectangles are refused before any output byte is written."""
    import os
    from meshflow_amd import _lib, synthetic
    from oracle import meshflow_oracle as mo
    F, H, W = 23, 72, 100
    frames = synthetic.frames_numpy(F, H, W, seed=4, kind='noise')
    ins = [f.copy() for f in frames]
    rect = (3, 5, 90, 66)
    want = np.stack(mo.crop_frames(list(frames), rect))
    pin = (ctypes.c_void_p * F)(*[f.ctypes.data for f in ins])
    old = {k: os.environ.get(k) for k in ('MF_PIPE_CHUNK', 'MF_PIPE_SLOTS')}
    try:
        for chunk, slots in ((None, None), (4, 2), (1, 3), (64, 8)):
            for k, v in (('MF_PIPE_CHUNK', chunk), ('MF_PIPE_SLOTS', slots)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            outs = [np.full((H, W, 3), 7, np.uint8) for _ in range(F)]
            pout = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs])
            ms = ctypes.c_float(-1)
            _lib.check(_lib.lib.mf_crop_resize_u8c3_host_frames(pin, pout, F, W, H, *rect, ctypes.byref(ms)))
            np.testing.assert_array_equal(np.stack(outs), want)
            assert ms.value > 0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    outs = [np.full((H, W, 3), 7, np.uint8) for _ in range(F)]
    pout = (ctypes.c_void_p * F)(*[f.ctypes.data for f in outs])
    for bad in ((50, 10, 40, 60), (0, 0, W, H - 1), (-1, 0, 10, 10), (0, 30, 10, 20)):
        assert _lib.lib.mf_crop_resize_u8c3_host_frames(pin, pout, F, W, H, *bad, None) == _lib.MF_ERR_INVALID_ARG
    assert all(int(o.min()) == 7 and int(o.max()) == 7 for o in outs)                   # untouched
    assert _lib.lib.mf_crop_resize_u8c3_host_frames(pin, pin, F, W, H, *rect, None) == _lib.MF_ERR_INVALID_ARG       # in place
    assert _lib.lib.mf_crop_resize_u8c3_host_frames(None, pout, F, W, H, *rect, None) == _lib.MF_ERR_INVALID_ARG


def test_2000_frame_1080p_clip_in_two_gigabytes_of_device_memory(dev):
    """A 2,000-frame 1080p clip (12.4 GB each way) host-to-host through `stabilize_clip(crop=True)`: the device-memory high-water mark
    stays below 2 GB (the ring: 8 slots x 2 directions x 16 frames; round 4 held 2 x 12.4 GB), the rectangle equals the resident
    pipeline's, sampled chunks equal the device operators on the same frames byte for byte, and an integer global shift gives the
    analytic answer on every frame."""
    import torch
    from meshflow_amd import _lib, synthetic
    from meshflow_amd.stabilizer import MeshFlowStabilizer
    import psutil
    F, H, W, R, C = 2000, 1080, 1920, 16, 16
    if psutil.virtual_memory().available < 18 * 2**30:
        pytest.skip('needs ~13 GB of host memory for the 2,000 cropped output frames')
    base = synthetic.frames_torch(40, H, W, dev, seed=2, kind='pattern').cpu().numpy()
    frames = [base[i % 40] for i in range(F)]                     # 40 distinct frames, cycled (input frames may repeat; outputs may not)
    disp, hom = synthetic.motion(F, R, C, seed=2)
    s = MeshFlowStabilizer(mesh_row_count=R, mesh_col_count=C, device='cuda:0')
    _lib.lib.mf_host_cache_release()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    before = _hbm_used()
    peak = [before]
    stop = [False]

    def watch():
        import time
        while not stop[0]:
            peak[0] = max(peak[0], _hbm_used())
            time.sleep(0.005)
    import threading
    th = threading.Thread(target=watch)
    th.start()
    try:
        # (keep_uncropped=False: ONE 12.4 GB output list -- the cropped frames -- instead of two, so that the test fits any box)
        out, bounds, stab, score, cropped = s.stabilize_clip(frames, disp, hom, crop=True, keep_uncropped=False)
    finally:
        stop[0] = True
        th.join()
    grown = peak[0] - before
    print(f'2000-frame clip: device memory high-water mark +{grown / 2**30:.2f} GiB')
    assert grown <= 2 * 2**30, grown
    assert out is None and len(cropped) == F
    # sampled chunks against the device operators on the same frames
    rect = tuple(int(v) for v in bounds)
    for i0 in (0, 16 * 61 + 3, F - 16):
        d_in = torch.from_numpy(np.stack(frames[i0:i0 + 16])).to(dev)
        d_out, crop_rows, _, d_cropped = _device_warp_crop(d_in, disp[i0:i0 + 16], stab[i0:i0 + 16], R, C, rect)
        assert np.array_equal(np.stack(cropped[i0:i0 + 16]), d_cropped.cpu().numpy())
    # the rectangle: every frame's rows through the crop scan of the whole clip's table, in pieces
    from meshflow_amd import ops
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    lo = [0, 0, W - 1, H - 1]
    for i0 in range(0, F, 250):
        table = ops.cell_table(t(disp[i0:i0 + 250]), t(stab[i0:i0 + 250]), W, H, R, C)
        rows = ops.crop_scan(table).cpu().numpy()
        lo = [max(lo[0], rows[:, 0].max()), max(lo[1], rows[:, 1].max()), min(lo[2], rows[:, 2].min()), min(lo[3], rows[:, 3].min())]
    assert rect == tuple(int(v) for v in lo)
    del cropped


def test_full_config4_clip_through_one_gpu(dev):
    """BASELINE config 4 WHOLE -- 1,200 frames of 3840 x 2160 (29.9 GB each way) -- host to host through ONE GPU's ring (4-frame chunks,
    1.6 GB of device memory), as the EIGHT frame-range shards of 150 frames the eight ranks of the real run take (host.shard_range), one
    after the other into ONE reused 3.7 GB output stack: 4.4 GB of host memory instead of 60, so the test runs on any box.  Motion =
    an integer global shift per frame, so every frame has the analytic answer: interior pixels moved by exactly that shift, the
    uncovered band in the border colour; every frame of every shard is checked against it, sampled frames also against the device
    operators, and the clip-level rectangle folded over the shards (what the 16-byte all-reduce does, mfs.py:1103-1106) against the
    crop scan of the whole clip's table."""
    import time
    import psutil
    import torch
    from meshflow_amd import _lib, host, ops, synthetic
    F, H, W, R, C, G = 1200, 2160, 3840, 16, 16, 8
    if psutil.virtual_memory().available < 8 * 2**30:
        pytest.skip('needs ~4.4 GB of host memory: one 150-frame output stack and 24 input frames')
    base = synthetic.frames_torch(24, H, W, dev, seed=7, kind='pattern').cpu().numpy()          # 24 distinct input frames, cycled
    disp, hom = synthetic.motion(F, R, C, seed=4)
    unstab = np.ascontiguousarray(disp)
    shift = np.zeros((F, 2))
    shift[:, 0] = (np.arange(F) % 7) - 3                                        # dx in -3..3
    shift[:, 1] = (np.arange(F) % 5) - 2                                        # dy in -2..2
    stab = np.ascontiguousarray(unstab + shift[:, None, None, :])               # content moves by (dx, dy)
    per = -(-F // G)
    out = np.empty((per, H, W, 3), np.uint8)
    fb = H * W * 3
    crop = np.zeros((F, 4), np.int32)
    border = (ctypes.c_uint8 * 3)(0, 0, 255)
    red = np.array([0, 0, 255], np.uint8)
    _lib.lib.mf_host_cache_release()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    before = _hbm_used()
    grown, dt, checked = 0, 0.0, 0
    for g in range(G):
        lo, hi = host.shard_range(F, G, g)
        n = hi - lo
        pin = (ctypes.c_void_p * n)(*[base[i % 24].ctypes.data for i in range(lo, hi)])
        pout = (ctypes.c_void_p * n)(*[out.ctypes.data + i * fb for i in range(n)])
        out[:n, ::64, ::64] = 7                                                  # (stale bytes of the previous shard must not pass for results)
        t0 = time.perf_counter()
        _lib.check(_lib.lib.mf_warp_u8c3_host_frames(pin, pout, _p(unstab[lo:hi]), _p(stab[lo:hi]), n, W, H, R, C, border, _p(crop[lo:hi]), None))
        dt += time.perf_counter() - t0
        grown = max(grown, _hbm_used() - before)
        for f in range(lo, hi):                                                  # EVERY frame against the analytic answer
            dx, dy = int(shift[f, 0]), int(shift[f, 1])
            src, got = base[f % 24], out[f - lo]
            ys, xs = slice(max(dy, 0) + 2, H + min(dy, 0) - 2), slice(max(dx, 0) + 2, W + min(dx, 0) - 2)
            # (row and column samples of the interior on every frame, the whole interior on every 29th: 1,200 full 4K compares are a minute of host time)
            if f % 29 == 0:
                assert np.array_equal(got[ys, xs], src[ys.start - dy:ys.stop - dy, xs.start - dx:xs.stop - dx]), f
            else:
                assert np.array_equal(got[ys, xs][::16, ::8], src[ys.start - dy:ys.stop - dy, xs.start - dx:xs.stop - dx][::16, ::8]), f
            if dx > 1:
                assert (got[:, :dx - 1] == red).all(), f                         # the uncovered band, in the border colour
            checked += 1
        for f in (lo, lo + 77):                                                  # and byte for byte against the device operators
            if f % 3 == 0 or f == lo:
                d_out, crop_rows, _, _ = _device_warp_crop(torch.from_numpy(base[f % 24][None]).to(dev), unstab[f:f + 1], stab[f:f + 1], R, C, (0, 0, W - 1, H - 1))
                assert np.array_equal(out[f - lo], d_out[0].cpu().numpy())
                assert np.array_equal(crop[f], crop_rows[0])
    assert checked == F
    print(f'full config 4 through one GPU: {dt:.2f} s = {F / dt:.0f} frames/s, {2 * F * fb / dt / 1e9:.1f} GB/s both ways, ring {grown / 2**30:.2f} GiB')
    assert grown <= 2 * 2**30
    # the clip-level rectangle: the shards' rows folded together = the crop scan of the whole clip's table
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rows = ops.crop_scan(ops.cell_table(t(unstab), t(stab), W, H, R, C)).cpu().numpy()
    assert np.array_equal(rows, crop)
    assert (crop[:, 0].max(), crop[:, 1].max(), crop[:, 2].min(), crop[:, 3].min()) == (rows[:, 0].max(), rows[:, 1].max(), rows[:, 2].min(), rows[:, 3].min())
    del out

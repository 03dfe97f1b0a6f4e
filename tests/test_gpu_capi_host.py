"""-m gpu: the host-buffer and multi-GPU entry points of the C ABI, raw ctypes (no torch in the calls).

mf_warp_u8c3_host / mf_warp_u8c3_host_frames: chunked, overlapped PCIe staging below Python (csrc/hostpipe.hip) -- several
chunks, ragged last chunk, cache reuse across calls and shapes, separate per-frame allocations, pinned buffers, errors.
mf_comm_init_all / mf_allreduce_crop / mf_gather_frames / mf_comm_destroy: RCCL directly (csrc/comm.hip); on the one-GPU
test box the communicator has one rank (the collectives then run device-to-device on that GPU)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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

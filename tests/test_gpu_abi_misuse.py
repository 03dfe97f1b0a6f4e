"""Invalid scalar arguments through the raw C ABI with valid (large enough) device buffers: every call comes back with an error code or
succeeds -- never a GPU fault, never a hang (3,800 calls: negative, zero and out-of-range frame counts, sizes, meshes, radii, sweep
counts and rectangles into the sweep, the cell table, the warp, the scan, crop + resize, the rectangle reduction and the score)."""
import ctypes
import itertools

import pytest

pytestmark = pytest.mark.gpu


def test_invalid_scalars_never_fault():
    import torch
    from meshflow_amd import _lib
    L = _lib.lib
    dev = torch.device('cuda:0')
    a, b, c = (torch.zeros(64 << 20, dtype=torch.uint8, device=dev) for _ in range(3))
    p, q, r = a.data_ptr(), b.data_ptr(), c.data_ptr()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    border = (ctypes.c_uint8 * 3)(1, 2, 3)
    calls = ok = 0

    def call(name, *args):
        nonlocal calls, ok
        rc = getattr(L, name)(*args)
        calls += 1
        ok += rc == 0
        if rc != 0:
            assert _lib.lib.mf_last_error()                         # (a message comes with every refusal)

    for F, S, omega, iters in itertools.product((-1, 0, 1, 5), (-1, 0, 3), (-1, 0, 1, 2147483647), (-1, 0, 2)):
        call('mf_jacobi_f64', p, q, r, r, r, F, S, omega, iters, st)
    for n, W, H, R, C in itertools.product((-1, 0, 1), (-1, 0, 1, 2, 8, 32768), (-1, 1, 2, 8, 32768), (-1, 0, 1, 65), (0, 1, 65)):
        if W > 0 and H > 0 and W * H * 3 * max(n, 1) > (32 << 20):
            continue
        call('mf_cell_table_f64', p, p, n, W, H, R, C, q, r, r, st)
        call('mf_warp_u8c3', p, q, r, n, W, H, R, C, border, r, st)
        call('mf_crop_scan_f64', q, n, W, H, R, C, r, st)
    for n, W, H in itertools.product((-1, 0, 1), (-1, 0, 1, 5, 32768), (-1, 0, 1, 5, 32768)):
        for rect in ((0, 0, 0, 0), (-1, 0, 3, 3), (2, 2, 1, 1), (0, 0, W, H), (0, 0, max(W, 1) - 1, max(H, 1) - 1), (2147483647, 0, 2147483647, 0)):
            if W > 0 and H > 0 and W * H * 3 > (32 << 20):
                continue
            call('mf_crop_resize_u8c3', p, q, n, W, H, *rect, r, st)
        call('mf_crop_reduce', r, n, W, H, q, st)
    for F, S in itertools.product((-1, 0, 1, 2, 3), (-1, 0, 1, 2, 3)):
        call('mf_stability_score_f64', p, F, S, q, r, st)
    torch.cuda.synchronize()                                        # a fault would surface here at the latest
    assert calls > 3500 and 0 < ok < 200
    probe = torch.arange(8, device=dev)
    assert int(probe.sum().item()) == 28                            # the device still answers

"""Device-level operators: torch tensors in, torch tensors out, HIP kernels underneath.

torch is used only for device memory, streams and (in `dist.py`) torch.distributed; every operator
hands raw device pointers to libmeshflow_hip.so through the C ABI (`_lib.py`) on torch's current
stream, so torch.cuda events and synchronisation see the kernels."""
import ctypes

import numpy as np
import torch

from . import _lib

_lib_ = _lib.lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _need(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f'{name} must be a CUDA/HIP torch tensor')
    if t.dtype != dtype:
        raise ValueError(f'{name} must have dtype {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')


def jacobi(b, taps, lam, inv_on, omega, iters, out=None):
    """`iters` Jacobi sweeps for all S series at once (mfs.py:844-878 x every vertex).
    b: (F, S) float64 device tensor, frame-major.  Returns x (F, S)."""
    _need(b, torch.float64, 'b')
    for name, t in (('taps', taps), ('lam', lam), ('inv_on', inv_on)):
        _need(t, torch.float64, name)
    F, S = b.shape
    if taps.numel() != 2 * omega + 1 or lam.numel() != F or inv_on.numel() != F:
        raise ValueError('coefficient sizes do not match (F, omega)')
    x = out if out is not None else torch.empty_like(b)
    _need(x, torch.float64, 'out')
    _lib.check(_lib_.mf_jacobi_f64(_ptr(b), _ptr(x), _ptr(taps), _ptr(lam), _ptr(inv_on), F, S, int(omega),
                                   int(iters), _stream()))
    return x


class CellTable:
    """Per-cell records + compact boxes of n frames (layout: include/meshflow_hip.h).  One object serves clips of any length of
    its geometry: `resize(n)` re-views the (grow-only) buffers for n frames."""

    def __init__(self, n, W, H, R, C, device):
        self.W, self.H, self.R, self.C = W, H, R, C
        self.device = device
        self.capacity = 0
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.bounds = None            # clip-level rectangle, filled by warp_clip
        self.resize(n)

    def resize(self, n):
        """View the table for n frames (allocates when n exceeds every earlier n; the contents do not survive a resize)."""
        if n > self.capacity:
            self.buf = torch.empty(_lib_.mf_cell_table_bytes(n, self.W, self.H, self.R, self.C), dtype=torch.uint8, device=self.device)
            self._crop = torch.empty((n, 4), dtype=torch.int32, device=self.device)
            self.capacity = n
        if n != getattr(self, 'n', None):
            self.n = n
            self.crop = self._crop[:n]
            off = _lib_.mf_cell_table_bounds_offset(n, self.W, self.H, self.R, self.C)
            # the rectangle as the kernels fold it together inside the table blob when no caller-owned tensor is given (valid after the
            # warp / crop scan of all n frames; overwritten by the next cell_table on this object)
            self.clip_bounds = self.buf[off:off + 16].view(torch.int32)
        return self

    def records(self):
        """(n, R*C, 32) float64 view of the records (for tests)."""
        nrec = self.n * self.R * self.C
        return self.buf[:nrec * _lib.CELL_DOUBLES * 8].view(torch.float64).view(self.n, self.R * self.C, _lib.CELL_DOUBLES)

    def check(self):
        """Raise if a cell had no homography (the reference would fail inside cv2.warpPerspective)."""
        bad = int(self.status.item())
        if bad:
            raise ValueError(f'{bad} degenerate mesh cell(s): no homography exists '
                             '(cv2.findHomography would return None)')


def _need_bounds(bounds):
    _need(bounds, torch.int32, 'bounds')
    if bounds.numel() != 4:
        raise ValueError('bounds must hold 4 int32 {left, top, right, bottom}')


def cell_table(unstab, stab, W, H, R, C, table=None, reset_status=True, bounds=None):
    """Per-cell homographies of n frames (mfs.py:1039-1048).  unstab/stab: (n, R+1, C+1, 2) or (n, V*2)
    float64 device tensors.  Also resets the per-frame crop values to their defaults (mfs.py:992-995).
    bounds: a caller-owned int32[4] device tensor that receives the clip-level rectangle (defaults here; `warp` / `crop_scan` called
    with the same tensor fold their frames into it) instead of the four words inside the table (`table.clip_bounds`)."""
    _need(unstab, torch.float64, 'unstab')
    _need(stab, torch.float64, 'stab')
    n = unstab.shape[0]
    V2 = (R + 1) * (C + 1) * 2
    if unstab.numel() != n * V2 or stab.numel() != n * V2:
        raise ValueError('displacement tensors do not match (n, R+1, C+1, 2)')
    if table is None:
        table = CellTable(n, W, H, R, C, unstab.device)
    else:
        if (table.W, table.H, table.R, table.C) != (W, H, R, C):
            raise ValueError('the cell table was made for another frame size / mesh')
        table.resize(n)
        if reset_status:
            table.status.zero_()          # reset_status=False: keep accumulating; the caller checks once later
    if bounds is None:
        _lib.check(_lib_.mf_cell_table_f64(_ptr(unstab), _ptr(stab), n, W, H, R, C, _ptr(table.buf), _ptr(table.crop),
                                           _ptr(table.status), _stream()))
    else:
        _need_bounds(bounds)
        _lib.check(_lib_.mf_cell_table_bounds_f64(_ptr(unstab), _ptr(stab), n, W, H, R, C, _ptr(table.buf), _ptr(table.crop),
                                                  _ptr(table.status), _ptr(bounds), _stream()))
    return table


def warp(frames, table, border_bgr=(0, 0, 255), out=None, bounds=None):
    """Mesh warp + crop scan of n frames (mfs.py:1000-1100).  frames: (n, H, W, 3) uint8 device tensor.
    Returns the stabilized frames; per-frame crop values accumulate in table.crop, the clip-level rectangle in `bounds` (the tensor
    `cell_table` was given) or, without one, in table.clip_bounds."""
    _need(frames, torch.uint8, 'frames')
    n, H, W, ch = frames.shape
    if ch != 3 or (n, W, H) != (table.n, table.W, table.H):
        raise ValueError('frames do not match the cell table (n, H, W, 3)')
    if out is None:
        out = torch.empty_like(frames)
    _need(out, torch.uint8, 'out')
    border = (ctypes.c_uint8 * 3)(*[int(np.clip(round(float(v)), 0, 255)) for v in border_bgr[:3]])
    if bounds is None:
        _lib.check(_lib_.mf_warp_u8c3(_ptr(frames), _ptr(out), _ptr(table.buf), n, W, H, table.R, table.C, border,
                                      _ptr(table.crop), _stream()))
    else:
        _need_bounds(bounds)
        _lib.check(_lib_.mf_warp_bounds_u8c3(_ptr(frames), _ptr(out), _ptr(table.buf), n, W, H, table.R, table.C, border,
                                             _ptr(table.crop), _ptr(bounds), _stream()))
    return out


def warp_clip(frames, unstab, stab, table, border_bgr=(0, 0, 255), out=None, chunks=4, prep_stream=None, bounds=None):
    """mfs.py:909-1108 for a clip resident in HBM as ONE call overlapped inside the clip (csrc/clippipe.hip): cell table + plan +
    crop scan + clip rectangle on `prep_stream` (a torch stream; None = the library's own, forked from the current stream), the warp
    of `chunks` frame ranges on torch's current stream, each waiting for its own table only.  chunks=0: in order on the current
    stream -- table, warp alone, rectangle (early on `prep_stream` when one is given).  Returns (stabilized frames,
    bounds): bounds = int32 {left, top, right, bottom} of the clip -- the caller's tensor when one is given, else table.bounds --,
    folded together by the kernels, final on the prep stream right after the tables' crop scan (and on the current stream after
    the call); per-frame values in table.crop; table.status accumulates degenerate cells."""
    _need(frames, torch.uint8, 'frames')
    _need(unstab, torch.float64, 'unstab')
    _need(stab, torch.float64, 'stab')
    n, H, W, ch = frames.shape
    V2 = (table.R + 1) * (table.C + 1) * 2
    if ch != 3 or (n, W, H) != (table.n, table.W, table.H) or unstab.numel() != n * V2 or stab.numel() != n * V2:
        raise ValueError('frames / displacements do not match the cell table')
    if out is None:
        out = torch.empty_like(frames)
    _need(out, torch.uint8, 'out')
    if bounds is None:                  # (without a caller-owned tensor: one per table, rewritten by the next call on it)
        if table.bounds is None:
            table.bounds = torch.empty(4, dtype=torch.int32, device=frames.device)
        bounds = table.bounds
    else:
        _need_bounds(bounds)
    border = (ctypes.c_uint8 * 3)(*[int(np.clip(round(float(v)), 0, 255)) for v in border_bgr[:3]])
    prep = ctypes.c_void_p(prep_stream.cuda_stream) if prep_stream is not None else None
    _lib.check(_lib_.mf_warp_clip_u8c3(_ptr(frames), _ptr(out), _ptr(unstab), _ptr(stab), n, W, H, table.R, table.C, border,
                                       _ptr(table.buf), _ptr(table.crop), _ptr(bounds), _ptr(table.status), int(chunks), prep, _stream()))
    return out, bounds


def crop_scan(table, bounds=None):
    """The four edge scans of mfs.py:1075-1098 from the cell table alone (no frame is touched): fills table.crop exactly as
    `warp` would (and folds the clip-level rectangle into `bounds` / table.clip_bounds).  Returns table.crop, (n, 4) int32
    {left, top, right, bottom}."""
    if bounds is None:
        _lib.check(_lib_.mf_crop_scan_f64(_ptr(table.buf), table.n, table.W, table.H, table.R, table.C, _ptr(table.crop), _stream()))
    else:
        _need_bounds(bounds)
        _lib.check(_lib_.mf_crop_scan_bounds_f64(_ptr(table.buf), table.n, table.W, table.H, table.R, table.C, _ptr(table.crop),
                                                 _ptr(bounds), _stream()))
    return table.crop


def crop_reduce(crop, W, H):
    """Clip-level bounds (mfs.py:1103-1106): int32 tensor {left, top, right, bottom}."""
    _need(crop, torch.int32, 'crop')
    bounds = torch.empty(4, dtype=torch.int32, device=crop.device)
    _lib.check(_lib_.mf_crop_reduce(_ptr(crop), crop.shape[0], W, H, _ptr(bounds), _stream()))
    return bounds


def crop_resize(frames, bounds, out=None):
    """Crop to the inclusive (left, top, right, bottom) and resize back to (W, H): mfs.py:1111-1157."""
    _need(frames, torch.uint8, 'frames')
    n, H, W, ch = frames.shape
    if ch != 3:
        raise ValueError('frames must be (n, H, W, 3)')
    left, top, right, bottom = (int(v) for v in bounds)
    if out is None:
        out = torch.empty_like(frames)
    _need(out, torch.uint8, 'out')
    work = torch.empty(_lib_.mf_crop_resize_workspace_bytes(W, H), dtype=torch.uint8, device=frames.device)
    _lib.check(_lib_.mf_crop_resize_u8c3(_ptr(frames), _ptr(out), n, W, H, left, top, right, bottom, _ptr(work), _stream()))
    return out


def vertex_motion(early, late, offsets, homographies, max_per_pair, W, H, R, C, ellipse_rows, ellipse_cols):
    """Vertex velocities and their running sum from matched features (mfs.py:236-452 after the tracker).
    early/late: (K_total, 2) float64 device tensors; offsets: (P+1,) int32; homographies: (P, 3, 3) float64.
    Returns (displacements (P+1, R+1, C+1, 2) float64, velocities (P, R+1, C+1, 2) float32, status (1,) int32);
    status != 0 means the reference's math.sqrt would have raised (mfs.py:444) -- see `vertex_motion_check`."""
    _need(early, torch.float64, 'early')
    _need(late, torch.float64, 'late')
    _need(offsets, torch.int32, 'offsets')
    _need(homographies, torch.float64, 'homographies')
    P = offsets.numel() - 1
    K = early.shape[0]
    if early.shape != late.shape or early.dim() != 2 or early.shape[1] != 2 or P < 0 or homographies.numel() != 9 * P:
        raise ValueError('features must be (K, 2) pairs with (P+1,) offsets and (P, 3, 3) homographies')
    dev = early.device
    vel = torch.empty((P, R + 1, C + 1, 2), dtype=torch.float32, device=dev)
    disp = torch.empty((P + 1, R + 1, C + 1, 2), dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(_lib_.mf_vertex_motion_workspace_bytes(K, int(max_per_pair), P, R, C), dtype=torch.uint8, device=dev)
    _lib.check(_lib_.mf_vertex_motion_f64(_ptr(early), _ptr(late), _ptr(offsets), _ptr(homographies), P, K,
                                          int(max_per_pair), W, H, R, C, int(ellipse_rows), int(ellipse_cols),
                                          _ptr(vel), _ptr(disp), _ptr(work), _ptr(status), _stream()))
    return disp, vel, status


def vertex_motion_check(status):
    if int(status.item()):
        raise ValueError('math domain error')          # what math.sqrt raises at mfs.py:444


def stability_score(stab):
    """Stability score (mfs.py:1216-1259) of device-resident paths: (F, R+1, C+1, 2) or (F, S) float64 tensor ->
    (score (1,) float64 tensor, per-series fractions (S,))."""
    _need(stab, torch.float64, 'stab')
    F = stab.shape[0]
    S = stab.numel() // F
    series = torch.empty(S, dtype=torch.float64, device=stab.device)
    score = torch.empty(1, dtype=torch.float64, device=stab.device)
    _lib.check(_lib_.mf_stability_score_f64(_ptr(stab), F, S, _ptr(series), _ptr(score), _stream()))
    return score, series

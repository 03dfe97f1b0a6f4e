"""CPU oracle for the vertex-motion accumulation that feeds the hot path.  TEST INFRASTRUCTURE ONLY.

SURVEY.md §8(f) row 3: the part of the reference's front-end (`/root/reference/meshflowstabilizer.py`,
``mfs.py``) that turns matched features + a global homography per frame pair into the per-vertex
displacement tensor the Jacobi sweep consumes:

  * ``_get_vertex_nearby_feature_residual_velocities``  mfs.py:365-452   ellipse splat
  * ``_get_unstabilized_vertex_velocities``             mfs.py:287-362   medians, global motion, 3x3 median blur
  * ``_get_unstabilized_vertex_displacements_and_homographies`` mfs.py:236-284   running sum over frames

Feature detection / tracking / RANSAC (mfs.py:455-629) stay outside: this file starts from their outputs.

dtype note.  The reference documents the feature arrays as CV_32FC2, but mfs.py:578 adds a Python list
(the sub-frame offset) to them, which promotes them to float64; every scalar taken from them at
mfs.py:425 is therefore a float64 and the ellipse arithmetic, the residual velocities and the medians are
float64.  This restatement (and the HIP kernel) take that path: feature arrays of any float dtype are
converted to float64 first (exactly).

Pinning status: the splat, the medians, the dtype flow and the running sum are PINNED -- the reference's own
methods are executed by ``oracle/gen_golden.py`` (stub ``cv2``) and their outputs are committed under
``tests/golden/motion_*.npz``.  The two OpenCV calls inside (``cv2.perspectiveTransform`` mfs.py:325,420 and
``cv2.medianBlur`` mfs.py:359-360) are restated from OpenCV 4.x (core/matmul.simd.hpp, imgproc/median_blur) and,
like the warp, are unpinned: the stub hands the reference these restatements.
"""
import math

import numpy as np

from .meshflow_oracle import FLT_EPSILON, perspective_transform_f32, vertex_x_y


def perspective_transform_f64(points_xy, H):
    """cv2.perspectiveTransform on float64 2-channel points (perspectiveTransform_64f): same formula as the
    float32 flavour, no final narrowing."""
    m = np.asarray(H, dtype=np.float64).reshape(9)
    p = np.asarray(points_xy, dtype=np.float64)
    x, y = p[..., 0], p[..., 1]
    w = (x * m[6] + y * m[7]) + m[8]
    ok = np.abs(w) > FLT_EPSILON
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        iw = 1.0 / w
        u = ((x * m[0] + y * m[1]) + m[2]) * iw
        v = ((x * m[3] + y * m[4]) + m[5]) * iw
    out = np.empty(p.shape, dtype=np.float64)
    out[..., 0] = np.where(ok, u, 0.0)
    out[..., 1] = np.where(ok, v, 0.0)
    return out


def feature_residual_velocities(early, late, H):
    """mfs.py:420: late - perspectiveTransform(early, H), float64 (K, 2)."""
    e = np.asarray(early, dtype=np.float64).reshape(-1, 2)
    l = np.asarray(late, dtype=np.float64).reshape(-1, 2)
    return l - perspective_transform_f64(e, H)


def ellipse_cover(feature_x, feature_y, frame_width, frame_height, mesh_rows, mesh_cols, ell_rows, ell_cols):
    """Vertices (row, col) inside the ellipse drawn around one feature, mfs.py:426-448, as a list of
    (row, first_col, last_col) with inclusive column ranges.  All arithmetic in float64, one rounding per
    operation, in the reference's order.  Raises ValueError where ``math.sqrt`` would (mfs.py:444)."""
    frow = (feature_y / frame_height) * mesh_rows                       # mfs.py:426
    fcol = (feature_x / frame_width) * mesh_cols                        # mfs.py:427
    half_rows = ell_rows / 2
    top = max(0, math.ceil(frow - half_rows))                           # mfs.py:438
    bottom = min(mesh_rows, math.floor(frow + half_rows))               # mfs.py:439 (inclusive here)
    spans = []
    for r in range(top, bottom + 1):
        q = (r - frow) / ell_rows
        half_width = ell_cols * math.sqrt(0.25 - q * q)                 # mfs.py:444
        left = max(0, math.ceil(fcol - half_width))                     # mfs.py:445
        right = min(mesh_cols, math.floor(fcol + half_width))           # mfs.py:446 (inclusive here)
        if left <= right:
            spans.append((r, left, right))
    return spans


def vertex_nearby_feature_residual_velocities(frame_width, frame_height, mesh_rows, mesh_cols, ell_rows, ell_cols,
                                              early, late, H):
    """mfs.py:365-452.  Returns two (R+1) x (C+1) nested lists of Python floats (x and y residual velocities
    of the features whose ellipse covers the vertex, in feature order)."""
    vx = [[[] for _ in range(mesh_cols + 1)] for _ in range(mesh_rows + 1)]
    vy = [[[] for _ in range(mesh_cols + 1)] for _ in range(mesh_rows + 1)]
    if early is None:
        return vx, vy
    e = np.asarray(early, dtype=np.float64).reshape(-1, 2)
    res = feature_residual_velocities(e, late, H)
    for k in range(e.shape[0]):
        rx, ry = float(res[k, 0]), float(res[k, 1])
        for r, left, right in ellipse_cover(float(e[k, 0]), float(e[k, 1]), frame_width, frame_height,
                                            mesh_rows, mesh_cols, ell_rows, ell_cols):
            for c in range(left, right + 1):
                vx[r][c].append(rx)
                vy[r][c].append(ry)
    return vx, vy


def median_or_zero(values):
    """``statistics.median(values) if values else 0`` (mfs.py:340-341): middle element, or the float64 mean of
    the two middle elements."""
    n = len(values)
    if n == 0:
        return 0.0
    s = sorted(values)
    return s[n // 2] if n % 2 else (s[n // 2 - 1] + s[n // 2]) / 2


def median_blur3_f32(img):
    """cv2.medianBlur(img float32 2-D, 3) (mfs.py:359-360): median of the 3x3 neighbourhood, borders replicated."""
    a = np.asarray(img, dtype=np.float32)
    p = np.pad(a, 1, mode='edge')
    h, w = a.shape
    stack = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)])
    return np.sort(stack, axis=0)[4].astype(np.float32)


def unstabilized_vertex_velocities(frame_width, frame_height, mesh_rows, mesh_cols, ell_rows, ell_cols,
                                   early, late, H, smooth=True):
    """mfs.py:287-362 from the matched features on: float32 (R+1, C+1, 2) vertex velocities."""
    grid = vertex_x_y(frame_width, frame_height, mesh_rows, mesh_cols)                       # (V, 1, 2) float32
    glob = (perspective_transform_f32(grid, H) - grid).reshape(mesh_rows + 1, mesh_cols + 1, 2)   # mfs.py:325-326
    lx, ly = vertex_nearby_feature_residual_velocities(frame_width, frame_height, mesh_rows, mesh_cols,
                                                       ell_rows, ell_cols, early, late, H)
    rx = np.array([[median_or_zero(v) for v in row] for row in lx], dtype=np.float64)        # mfs.py:338-353
    ry = np.array([[median_or_zero(v) for v in row] for row in ly], dtype=np.float64)
    vx = (glob[:, :, 0].astype(np.float64) + rx).astype(np.float32)                          # mfs.py:354-355
    vy = (glob[:, :, 1].astype(np.float64) + ry).astype(np.float32)
    if smooth:
        vx, vy = median_blur3_f32(vx), median_blur3_f32(vy)                                  # mfs.py:359-360
    return np.dstack((vx, vy))


def unstabilized_vertex_displacements(frame_width, frame_height, mesh_rows, mesh_cols, ell_rows, ell_cols,
                                      features_by_pair, homographies):
    """mfs.py:268-282: displacements[0] = 0, displacements[t+1] = displacements[t] + velocity_t (float64).
    ``features_by_pair[t]`` = (early, late) for frames t, t+1; ``homographies[t]`` the matching 3x3."""
    num_frames = len(features_by_pair) + 1
    disp = np.zeros((num_frames, mesh_rows + 1, mesh_cols + 1, 2))
    vel = np.zeros((num_frames - 1, mesh_rows + 1, mesh_cols + 1, 2), dtype=np.float32)
    for t, (early, late) in enumerate(features_by_pair):
        vel[t] = unstabilized_vertex_velocities(frame_width, frame_height, mesh_rows, mesh_cols, ell_rows, ell_cols,
                                                early, late, homographies[t])
        disp[t + 1] = disp[t] + vel[t]
    return disp, vel

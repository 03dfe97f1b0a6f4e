"""Host-side (NumPy, float64) pieces of the hot path that stay on the CPU because they are O(F) or
O(F*omega) scalar work: adaptive weights, Jacobi band coefficients, vertex grid, stability score.

`mfs.py:N` = /root/reference/meshflowstabilizer.py line N."""
import math

import numpy as np

ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL = 0
ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED = 1
ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH = 2
ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW = 3
ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH_VALUE = 100
ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW_VALUE = 1

VALID_DEFINITIONS = (0, 1, 2, 3)


def adaptive_weights(num_frames, frame_width, frame_height, definition, homographies):
    """lambda_t (mfs.py:786-841).  Same LAPACK eigenvalue routine as the reference, batched."""
    if definition in (ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL, ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED):
        affine = np.array(homographies, dtype=np.float64, copy=True)
        if affine.shape != (num_frames, 3, 3):
            raise ValueError(f'homographies must have shape ({num_frames}, 3, 3)')
        affine[:, 2, :] = [0, 0, 1]                                              # mfs.py:816
        mags = np.sort(np.abs(np.linalg.eigvals(affine)), axis=1)                # mfs.py:821
        tau = np.sqrt((affine[:, 0, 2] / frame_width) ** 2 + (affine[:, 1, 2] / frame_height) ** 2)
        a = mags[:, -2] / mags[:, -1]                                            # mfs.py:824
        c1 = -1.93 * tau + 0.95                                                  # mfs.py:826
        sign = 1.0 if definition == ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL else -1.0
        c2 = 5.83 * a + sign * 4.88                                              # mfs.py:829, 831
        return np.maximum(np.minimum(c1, c2), 0.0)                               # mfs.py:833-835
    if definition == ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH:
        return np.full((num_frames,), float(ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH_VALUE))
    if definition == ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW:
        return np.full((num_frames,), float(ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW_VALUE))
    raise ValueError('bad adaptive_weights_definition')


def jacobi_band_coefficients(num_frames, frame_width, frame_height, definition, homographies, omega):
    """(taps[2*omega+1], lam[F], inv_on[F]): the band of the reference's dense F x F system.

    mfs.py:713-783 builds off[t,r] = -2*lam_t*w(t-r) masked to |t-r| <= omega (the band INCLUDES the
    diagonal) and on[t] = 1 + 2*lam_t*sum_{r=0}^{F-1} w(t-r) over the WHOLE row (the sum is taken before
    the mask, mfs.py:775).  w(d) = exp(-((3/omega)*d)^2).  O(F) here instead of O(F^2)."""
    F = int(num_frames)
    omega = int(omega)
    if omega < 1:
        raise ValueError('temporal_smoothing_radius must be >= 1')
    d = np.arange(-omega, omega + 1)
    taps = np.exp(-np.square((3 / omega) * d))
    lam = np.asarray(adaptive_weights(F, frame_width, frame_height, definition, homographies), dtype=np.float64)
    # row sums: sum_{r=0}^{F-1} w(t-r) = sum_{d=t-F+1}^{t} w(d)
    dd = np.arange(-(F - 1), F)
    wfull = np.exp(-np.square((3 / omega) * dd))
    csum = np.concatenate([[0.0], np.cumsum(wfull)])
    t = np.arange(F)
    row = csum[t + F] - csum[t]                     # dd index of d is d + F - 1; range [t-F+1, t]
    on = 1 + 2 * lam * row
    return taps, lam, np.reciprocal(on)             # mfs.py:873


def vertex_x_y(frame_width, frame_height, mesh_rows, mesh_cols):
    """mfs.py:881-906: (V, 1, 2) float32 vertex pixel coordinates."""
    xs = [math.ceil((frame_width - 1) * (col / mesh_cols)) for col in range(mesh_cols + 1)]
    ys = [math.ceil((frame_height - 1) * (row / mesh_rows)) for row in range(mesh_rows + 1)]
    out = np.empty((mesh_rows + 1, mesh_cols + 1, 2), dtype=np.float32)
    out[..., 0] = np.asarray(xs, dtype=np.float32)[None, :]
    out[..., 1] = np.asarray(ys, dtype=np.float32)[:, None]
    return out.reshape(-1, 1, 2)


def stability_score(vertex_stabilized_displacements_by_frame_index):
    """mfs.py:1216-1259 (pure NumPy in the reference as well)."""
    xs, ys = np.swapaxes(vertex_stabilized_displacements_by_frame_index, 0, 3)
    parts = []
    for profiles in (np.diff(xs), np.diff(ys)):
        energy = np.square(np.abs(np.fft.fft(profiles)))
        total = np.sum(energy, axis=2)
        low = np.sum(energy[:, :, 1:6], axis=2)
        parts.append(np.mean(low / total))
    return (parts[0] + parts[1]) / 2.0


def shard_range(num_frames, world_size, rank):
    """Contiguous frame range [lo, hi) of `rank`: ceil(F/G) frames per rank, the tail ranks shorter."""
    per = -(-num_frames // world_size)
    lo = min(rank * per, num_frames)
    hi = min(lo + per, num_frames)
    return lo, hi


def mesh_cropping_ratio_and_distortion(frame_width, frame_height, mesh_rows, mesh_cols, unstab_disp, stab_disp,
                                       crop_boundaries):
    """(cropping_ratio, distortion_score) from the mesh itself -- a DEVIATION from mfs.py:1160-1212, not parity-pinned.

    The reference estimates, for every frame, a homography between the unstabilized and the cropped frame from
    FAST/LK feature matches (mfs.py:1195) and takes 1/(H00*H11) (mean over frames, mfs.py:1203, 1212) and the ratio
    of the two largest eigenvalue magnitudes of its affine part (MIN over frames, mfs.py:1206-1212).  The feature
    tracker is outside this build; here the same two formulas are applied to the least-squares homography through
    the (R+1)(C+1) exactly known vertex correspondences: unstabilized grid vertex -> its stabilized position ->
    crop + resize (pixel centres: x_c = (x_s - left + 0.5) * W/cw - 0.5)."""
    left, top, right, bottom = (float(v) for v in crop_boundaries)
    cw, ch = right - left + 1.0, bottom - top + 1.0
    grid = vertex_x_y(frame_width, frame_height, mesh_rows, mesh_cols).reshape(-1, 2).astype(np.float64)
    motion = (np.asarray(stab_disp, dtype=np.float64) - np.asarray(unstab_disp, dtype=np.float64)).reshape(len(stab_disp), -1, 2)
    ratios = np.empty(len(motion), dtype=np.float32)
    distortions = np.empty(len(motion), dtype=np.float32)
    X, Y = grid[:, 0], grid[:, 1]
    for f in range(len(motion)):
        ps = grid + motion[f]
        x = (ps[:, 0] - left + 0.5) * (frame_width / cw) - 0.5
        y = (ps[:, 1] - top + 0.5) * (frame_height / ch) - 0.5
        zeros, ones = np.zeros_like(X), np.ones_like(X)
        A = np.concatenate([np.stack([X, Y, ones, zeros, zeros, zeros, -x * X, -x * Y], axis=1),
                            np.stack([zeros, zeros, zeros, X, Y, ones, -y * X, -y * Y], axis=1)])
        h, *_ = np.linalg.lstsq(A, np.concatenate([x, y]), rcond=None)
        Hm = np.append(h, 1.0).reshape(3, 3)
        ratios[f] = 1 / (Hm[0][0] * Hm[1][1])                                             # mfs.py:1203
        affine = Hm.copy()
        affine[2] = [0, 0, 1]                                                            # mfs.py:1207
        mags = np.sort(np.abs(np.linalg.eigvals(affine)))
        distortions[f] = mags[-2] / mags[-1]                                              # mfs.py:1209
    return np.mean(ratios), np.min(distortions)                                           # mfs.py:1212


def pack_features(features_by_pair):
    """[(early, late), ...] as `_get_matched_features_and_homography` returns them (mfs.py:528: (K, 1, 2) arrays, or
    None when too few features were found) -> (early (K_total, 2) float64, late, offsets (P+1,) int32, max per pair).
    float32 inputs are widened exactly; the reference's own flow already carries float64 (mfs.py:578)."""
    early, late = [], []
    for pair in features_by_pair:
        e, l = pair[0], pair[1]
        e = np.zeros((0, 2)) if e is None else np.asarray(e, dtype=np.float64).reshape(-1, 2)
        l = np.zeros((0, 2)) if l is None else np.asarray(l, dtype=np.float64).reshape(-1, 2)
        if e.shape != l.shape:
            raise ValueError('early and late features of a pair must have the same shape')
        early.append(e)
        late.append(l)
    counts = [len(e) for e in early]
    offsets = np.cumsum([0] + counts).astype(np.int32)
    cat = lambda parts: np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros((0, 2))
    return cat(early), cat(late), offsets, max(counts, default=0)

"""CPU oracle (NumPy, float64) for the MeshFlow hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the arithmetic of the two hot methods of the reference
(`/root/reference/meshflowstabilizer.py`, cited below as ``mfs.py:N``):

  * ``_get_stabilized_vertex_displacements``  (mfs.py:632-710)  -- Jacobi temporal smoothing
  * ``_get_stabilized_frames_and_crop_boundaries`` (mfs.py:909-1108) -- per-cell mesh warp

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``meshflow_amd``) never imports anything from ``oracle/``.

Pinning status
--------------
* Jacobi half (coefficients, sweep, vertex grid, stability score): PINNED.  The reference is
  imported in the build container with a stub ``cv2`` module (``oracle/gen_golden.py``) and its
  outputs are committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file
  against them.
* Warp half: **parity unpinned**.  The arithmetic lives in OpenCV (``cv2``, version not pinned by
  the reference's requirements.txt:1-2, sources absent from /root/reference, not installed in
  this image).  The four cv2 calls on the path (``findHomography`` mfs.py:1041-1042,
  ``warpPerspective`` mfs.py:1052, ``perspectiveTransform`` mfs.py:1054, ``remap`` mfs.py:1063)
  are restated below from OpenCV 4.x's published algorithms (modules/calib3d/src/fundam.cpp,
  modules/imgproc/src/imgwarp.cpp, modules/core/src/matmul.simd.hpp, modules/core/src/lapack.cpp)
  and pinned only by known-answer tests (``tests/test_oracle_warp_kat.py``).
  MODELLED VERSION RANGE: **OpenCV 4.5 ... 4.10** -- the FIXED-POINT 8-bit kernels: ``remap`` /
  ``warpPerspective`` with INTER_BITS = 5 coordinates (``cvRound(x * 32)``), the 2^15-scaled weight
  table and ``(sum + 2^14) >> 15``; ``resize`` with 11-bit coefficients (``>> 4 ... >> 16 ... + 2 >> 2``);
  4-point ``findHomography`` = normalised DLT without refinement.  The reference's era (numpy 1.22 /
  tqdm 4.56, requirements.txt) is OpenCV 4.5.x.  OpenCV >= 4.11 blends 8-bit ``remap`` /
  ``warpPerspective`` with float weights: a last-bit difference against such a build is a version
  difference.  ``tests/test_cv2_crosscheck.py`` prints ``cv2.__version__``, asserts bit-exactness
  inside the range and BASELINE.json's bars (<= 1 LSB, 1e-4) outside it; ``tools/settle_parity.sh``
  is the one command to run where a ``cv2`` exists.

The warp is written here in the reference's own shape: one full-frame pass per mesh cell in
row-major order, later cells painting over earlier ones (mfs.py:1031-1061), then one remap
(mfs.py:1063-1069) and four edge scans (mfs.py:1075-1098).  It is O(R*C*H*W) per frame and only
meant for small frames; ``oracle/warp_oracle.c`` is the single-pass restatement of the same
arithmetic that is used at larger sizes, and is itself checked against this file.
"""
import math

import numpy as np

ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL = 0       # mfs.py:32
ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED = 1        # mfs.py:33
ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH = 2  # mfs.py:34
ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW = 3   # mfs.py:35
CONSTANT_HIGH_VALUE = 100                      # mfs.py:39
CONSTANT_LOW_VALUE = 1                         # mfs.py:40

FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_EPSILON = float(np.finfo(np.float64).eps)
INT_MIN = -2147483648
INT_MAX = 2147483647


# ------------------------------------------------------------------------------------------------
# Jacobi half
# ------------------------------------------------------------------------------------------------

def adaptive_weights(num_frames, frame_width, frame_height, definition, homographies):
    """lambda_t per frame.  Follows mfs.py:786-841 statement by statement."""
    if definition in (ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL, ADAPTIVE_WEIGHTS_DEFINITION_FLIPPED):
        affine = np.array(homographies, dtype=np.float64, copy=True)   # mfs.py:815
        affine[:, 2, :] = [0, 0, 1]                                    # mfs.py:816
        lam = np.empty((num_frames,))
        for t in range(num_frames):
            h = affine[t]
            e = np.sort(np.abs(np.linalg.eigvals(h)))                 # mfs.py:821
            tau = math.sqrt((h[0, 2] / frame_width) ** 2 + (h[1, 2] / frame_height) ** 2)  # mfs.py:823
            a = e[-2] / e[-1]                                          # mfs.py:824
            c1 = -1.93 * tau + 0.95                                    # mfs.py:826
            if definition == ADAPTIVE_WEIGHTS_DEFINITION_ORIGINAL:
                c2 = 5.83 * a + 4.88                                   # mfs.py:829
            else:
                c2 = 5.83 * a - 4.88                                   # mfs.py:831
            lam[t] = max(min(c1, c2), 0)                               # mfs.py:833-835
        return lam
    if definition == ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_HIGH:
        return np.full((num_frames,), CONSTANT_HIGH_VALUE)             # mfs.py:837 (int array)
    if definition == ADAPTIVE_WEIGHTS_DEFINITION_CONSTANT_LOW:
        return np.full((num_frames,), CONSTANT_LOW_VALUE)              # mfs.py:839
    raise ValueError('bad adaptive_weights_definition')


def jacobi_method_input(num_frames, frame_width, frame_height, definition, homographies, omega):
    """Dense (off, on) exactly as mfs.py:713-783 builds them (O(F^2); small F only)."""
    rows, cols = np.indices((num_frames, num_frames))                  # mfs.py:745
    w = np.exp(-np.square((3 / omega) * (rows - cols)))                # mfs.py:750-752 (w[t,t] = 1)
    lam = adaptive_weights(num_frames, frame_width, frame_height, definition, homographies)
    comb = np.matmul(np.diag(lam), w)                                  # mfs.py:763
    off = -2 * comb                                                    # mfs.py:767
    on = 1 + 2 * np.sum(comb, axis=1)                                  # mfs.py:775 (full row, pre-mask)
    mask = np.zeros(off.shape)                                         # mfs.py:778-781
    for i in range(-omega, omega + 1):
        mask += np.diag(np.ones(num_frames - abs(i)), i)
    off = np.where(mask, off, 0)
    return off, on


def jacobi_method_output_dense(off, on, x_start, b, iters):
    """mfs.py:844-878: x <- diag(1/on) @ (b - off @ x), `iters` times, dense matmuls."""
    x = x_start.copy()
    rdiag = np.diag(np.reciprocal(on))
    for _ in range(iters):
        x = np.matmul(rdiag, b - np.matmul(off, x))
    return x


def jacobi_band_coefficients(num_frames, frame_width, frame_height, definition, homographies, omega):
    """O(F*omega) form of mfs.py:713-783: (taps[2*omega+1], lam[F], on[F]).

    taps[d+omega] = w_d = exp(-((3/omega)*d)^2); off[t, t+d] = -2*(lam_t*w_d) for |d| <= omega
    (the band INCLUDES d = 0, mfs.py:779); on[t] = 1 + 2*sum_{r=0}^{F-1} lam_t*w_{t-r} over the
    whole row, not only the band (mfs.py:775 runs before the mask of mfs.py:778-781).
    """
    d = np.arange(-omega, omega + 1)
    taps = np.exp(-np.square((3 / omega) * d))
    lam = np.asarray(adaptive_weights(num_frames, frame_width, frame_height, definition, homographies),
                     dtype=np.float64)
    t = np.arange(num_frames)
    full = np.exp(-np.square((3 / omega) * (t[:, None] - t[None, :])))
    on = 1 + 2 * np.sum(lam[:, None] * full, axis=1)
    return taps, lam, on


def jacobi_banded(b, taps, lam, on, omega, iters):
    """Banded restatement of mfs.py:871-878 for all series at once.

    b: (F, S) float64, S independent series (vertex x component).  Equivalent, up to summation
    order, to the reference's dense products: x_new[t] = (1/on[t]) * (b[t] - sum_d off[t,t+d] x[t+d])
    with off[t,t+d] = -2*(lam[t]*taps[d]) and out-of-range t+d dropped.
    """
    F, S = b.shape
    inv_on = np.reciprocal(on)                                         # mfs.py:873
    x = b.copy()                                                       # mfs.py:871 (x_start is b, mfs.py:699-703)
    for _ in range(iters):
        xp = np.zeros((F + 2 * omega, S))
        xp[omega:omega + F] = x
        acc = np.zeros((F, S))
        for k in range(2 * omega + 1):
            coef = -2 * (lam * taps[k])                                # off[t, t+d], d = k - omega
            acc += coef[:, None] * xp[k:k + F]
        x = inv_on[:, None] * (b - acc)
    return x


def stabilized_vertex_displacements(frame_width, frame_height, definition, unstab_disp, homographies,
                                    omega, iters, dense=False):
    """mfs.py:632-710 for every vertex (the reference's R != C index bug, mfs.py:696-697, is not kept)."""
    F = unstab_disp.shape[0]
    b = np.ascontiguousarray(unstab_disp, dtype=np.float64).reshape(F, -1)
    if dense:
        off, on = jacobi_method_input(F, frame_width, frame_height, definition, homographies, omega)
        x = np.empty_like(b)
        for s in range(0, b.shape[1], 2):                              # one vertex = two columns
            x[:, s:s + 2] = jacobi_method_output_dense(off, on, b[:, s:s + 2], b[:, s:s + 2], iters)
    else:
        taps, lam, on = jacobi_band_coefficients(F, frame_width, frame_height, definition, homographies, omega)
        x = jacobi_banded(b, taps, lam, on, omega, iters)
    return x.reshape(unstab_disp.shape)


def stability_score(stab_disp):
    """mfs.py:1216-1259."""
    xs, ys = np.swapaxes(stab_disp, 0, 3)                              # mfs.py:1240
    scores = []
    for prof in (np.diff(xs), np.diff(ys)):                            # mfs.py:1241-1242
        e = np.square(np.abs(np.fft.fft(prof)))                        # mfs.py:1244-1245
        total = np.sum(e, axis=2)
        low = np.sum(e[:, :, 1:6], axis=2)                             # mfs.py:1250-1251
        scores.append(np.mean(low / total))
    return (scores[0] + scores[1]) / 2.0                               # mfs.py:1259


# ------------------------------------------------------------------------------------------------
# Warp half: helpers
# ------------------------------------------------------------------------------------------------

def vertex_x_y(frame_width, frame_height, mesh_rows, mesh_cols):
    """mfs.py:881-906: (V,1,2) float32, x = ceil((W-1)*(col/C)), y = ceil((H-1)*(row/R))."""
    return np.array([
        [[math.ceil((frame_width - 1) * (col / mesh_cols)),
          math.ceil((frame_height - 1) * (row / mesh_rows))]]
        for row in range(mesh_rows + 1)
        for col in range(mesh_cols + 1)
    ], dtype=np.float32)


def _matmul3(a, b):
    """3x3 product with each entry summed k = 0,1,2 in order (OpenCV small-matrix gemm)."""
    c = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            s = a[i][0] * b[0][j]
            s = s + a[i][1] * b[1][j]
            s = s + a[i][2] * b[2][j]
            c[i][j] = s
    return c


def solve8_partial_pivot(A, rhs):
    """8x8 Gaussian elimination, partial pivoting (first largest |pivot|), fixed operation order.

    This exact operation order is what `oracle/warp_oracle.c` and the HIP cell-table kernel follow,
    so all three agree bit for bit (no FMA contraction anywhere).
    """
    A = np.array(A, dtype=np.float64)
    r = np.array(rhs, dtype=np.float64)
    n = 8
    for k in range(n):
        p = k
        best = abs(A[k][k])
        for i in range(k + 1, n):
            if abs(A[i][k]) > best:
                best = abs(A[i][k])
                p = i
        if best == 0.0:
            return None
        if p != k:
            A[[k, p]] = A[[p, k]]
            r[[k, p]] = r[[p, k]]
        for i in range(k + 1, n):
            f = A[i][k] / A[k][k]
            for j in range(k + 1, n):
                A[i][j] = A[i][j] - f * A[k][j]
            r[i] = r[i] - f * r[k]
    h = np.zeros(n)
    for i in range(n - 1, -1, -1):
        s = r[i]
        for j in range(i + 1, n):
            s = s - A[i][j] * h[j]
        h[i] = s / A[i][i]
    return h


def _square_to_quad(p0, p1, p2, p3):
    """The homography that maps the unit square's corners (0,0), (1,0), (1,1), (0,1) onto p0, p1, p2, p3 (Heckbert, "Fundamentals of
    texture mapping and image warping", 1989, section 2.2.3: the 8 equations of the 4-point problem solved by hand), as the row-major
    [a, b, c, d, e, f, g, h, 1]; None when p1, p2, p3 are collinear.  Fixed operation order, shared with oracle/warp_oracle.c and the
    HIP cell-table kernel (no FMA contraction anywhere): all three agree bit for bit."""
    sx = ((p0[0] - p1[0]) + p2[0]) - p3[0]
    sy = ((p0[1] - p1[1]) + p2[1]) - p3[1]
    dx1 = p1[0] - p2[0]; dx2 = p3[0] - p2[0]
    dy1 = p1[1] - p2[1]; dy2 = p3[1] - p2[1]
    den = dx1 * dy2 - dx2 * dy1
    if not abs(den) > 1e-10:                 # p1, p2, p3 collinear (the caller normalises the points: coordinates of O(1))
        return None
    g = (sx * dy2 - dx2 * sy) / den
    h = (dx1 * sy - sx * dy1) / den
    return [(p1[0] - p0[0]) + g * p1[0], (p3[0] - p0[0]) + h * p3[0], p0[0],
            (p1[1] - p0[1]) + g * p1[1], (p3[1] - p0[1]) + h * p3[1], p0[1],
            g, h, 1.0]


def _adjugate3(S):
    """Adjugate of a row-major 3x3 whose last entry is 1 (S times its adjugate = det(S) * identity)."""
    a, b, c, d, e, f, g, h, _ = S
    return [[e - f * h, c * h - b, b * f - c * e],
            [f * g - d, a - c * g, c * d - a * f],
            [d * h - e * g, b * g - a * h, a * e - b * d]]


def find_homography_4pt(src_pts, dst_pts, solver='closed'):
    """cv2.findHomography(src, dst) for exactly 4 points, method 0 (mfs.py:1041-1042).

    Restates OpenCV's HomographyEstimatorCallback::runKernel (calib3d/fundam.cpp): both point sets
    are converted to float32; each is normalised to zero centroid and unit mean absolute deviation;
    the 8x9 DLT system is formed in normalised coordinates; its null vector is de-normalised as
    invHnorm * H0 * Hnorm2 and scaled by 1/H[2][2].  With 4 points there is no RANSAC and no LM
    refinement.

    OpenCV extracts the null vector as the eigenvector of the smallest eigenvalue of L^T L
    (Jacobi eigen-solver); its iteration is not reproducible bit-for-bit across builds, and with exactly 4 points the 8 equations
    have ONE solution, so any exact solver restates it to rounding.  `solver='closed'` (what the C oracle and the HIP kernel compute,
    since round 4) takes that solution in closed form -- square -> quad of the normalised destination points times the adjugate of
    square -> quad of the normalised source points (Heckbert): ~1,000 instructions per cell and direction on the GPU instead of ~3,500
    for an 8 x 8 elimination with predicated row exchanges; `solver='gauss'` (rounds 1-3) solves the 8 equations with h8 = 1 by Gaussian
    elimination with partial pivoting; `solver='eigh'` follows the L^T L route with numpy's symmetric eigen-solver.  The three agree to
    ~1e-12 relative (tests/test_oracle_warp_kat.py); the last two exist to cross-check the first.
    """
    M = np.asarray(src_pts, dtype=np.float64).reshape(4, 2).astype(np.float32).astype(np.float64)
    m = np.asarray(dst_pts, dtype=np.float64).reshape(4, 2).astype(np.float32).astype(np.float64)
    count = 4
    cM = [0.0, 0.0]
    cm = [0.0, 0.0]
    for i in range(count):
        cm[0] += m[i][0]; cm[1] += m[i][1]
        cM[0] += M[i][0]; cM[1] += M[i][1]
    cm = [cm[0] / count, cm[1] / count]
    cM = [cM[0] / count, cM[1] / count]
    sm = [0.0, 0.0]
    sM = [0.0, 0.0]
    for i in range(count):
        sm[0] += abs(m[i][0] - cm[0]); sm[1] += abs(m[i][1] - cm[1])
        sM[0] += abs(M[i][0] - cM[0]); sM[1] += abs(M[i][1] - cM[1])
    if min(abs(sm[0]), abs(sm[1]), abs(sM[0]), abs(sM[1])) < DBL_EPSILON:
        return None
    sm = [count / sm[0], count / sm[1]]
    sM = [count / sM[0], count / sM[1]]
    inv_h_norm = [[1.0 / sm[0], 0.0, cm[0]], [0.0, 1.0 / sm[1], cm[1]], [0.0, 0.0, 1.0]]
    h_norm2 = [[sM[0], 0.0, -cM[0] * sM[0]], [0.0, sM[1], -cM[1] * sM[1]], [0.0, 0.0, 1.0]]

    L = np.zeros((8, 9))
    for i in range(count):
        x = (m[i][0] - cm[0]) * sm[0]; y = (m[i][1] - cm[1]) * sm[1]
        X = (M[i][0] - cM[0]) * sM[0]; Y = (M[i][1] - cM[1]) * sM[1]
        L[2 * i] = [X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x]
        L[2 * i + 1] = [0, 0, 0, X, Y, 1, -y * X, -y * Y, -y]
    if solver == 'closed':
        # corner order: the reference hands the corners over as TL, TR, BL, BR (mfs.py:1039-1040); the unit square's cyclic order is
        # (0,0), (1,0), (1,1), (0,1) = points 0, 1, 3, 2 -- any labelling works as long as both sets use the same one
        nm = [((m[i][0] - cm[0]) * sm[0], (m[i][1] - cm[1]) * sm[1]) for i in range(count)]
        nM = [((M[i][0] - cM[0]) * sM[0], (M[i][1] - cM[1]) * sM[1]) for i in range(count)]
        s_dst = _square_to_quad(nm[0], nm[1], nm[3], nm[2])
        s_src = _square_to_quad(nM[0], nM[1], nM[3], nM[2])
        if s_dst is None or s_src is None:
            return None
        adj_src = _adjugate3(s_src)
        # either quad with ANY three corners collinear: its square -> quad map is singular (normalised coordinates are O(1); the same
        # operations, hence the same decision, as oracle/warp_oracle.c and csrc/cell_table.hip)
        det_s = (s_src[0] * adj_src[0][0] + s_src[1] * adj_src[1][0]) + s_src[2] * adj_src[2][0]
        det_d = (s_dst[0] * (s_dst[4] - s_dst[5] * s_dst[7]) + s_dst[1] * (s_dst[5] * s_dst[6] - s_dst[3])) + s_dst[2] * (s_dst[3] * s_dst[7] - s_dst[4] * s_dst[6])
        if not abs(det_s) > 1e-10 or not abs(det_d) > 1e-10:
            return None
        H0 = np.array(_matmul3([s_dst[0:3], s_dst[3:6], s_dst[6:9]], adj_src))
    elif solver == 'gauss':
        h = solve8_partial_pivot(L[:, :8], -L[:, 8])
        if h is None:
            return None
        H0 = np.array([[h[0], h[1], h[2]], [h[3], h[4], h[5]], [h[6], h[7], 1.0]])
    elif solver == 'eigh':
        _, vecs = np.linalg.eigh(L.T @ L)
        H0 = vecs[:, 0].reshape(3, 3)
    else:
        raise ValueError(solver)
    Htemp = _matmul3(inv_h_norm, H0)
    H = _matmul3(Htemp, h_norm2)
    if H[2][2] == 0.0:
        return None
    return H * (1.0 / H[2][2])


def invert3x3(S):
    """cv::invert for a 3x3 CV_64F matrix (core/lapack.cpp closed form, `det3` then adjugate*1/det)."""
    d = (S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1])
         - S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0])
         + S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]))
    if d == 0.0:
        return np.zeros((3, 3))
    d = 1.0 / d
    t = np.empty((3, 3))
    t[0][0] = (S[1][1] * S[2][2] - S[1][2] * S[2][1]) * d
    t[0][1] = (S[0][2] * S[2][1] - S[0][1] * S[2][2]) * d
    t[0][2] = (S[0][1] * S[1][2] - S[0][2] * S[1][1]) * d
    t[1][0] = (S[1][2] * S[2][0] - S[1][0] * S[2][2]) * d
    t[1][1] = (S[0][0] * S[2][2] - S[0][2] * S[2][0]) * d
    t[1][2] = (S[0][2] * S[1][0] - S[0][0] * S[1][2]) * d
    t[2][0] = (S[1][0] * S[2][1] - S[1][1] * S[2][0]) * d
    t[2][1] = (S[0][1] * S[2][0] - S[0][0] * S[2][1]) * d
    t[2][2] = (S[0][0] * S[1][1] - S[0][1] * S[1][0]) * d
    return t


def _cv_round_f64(v):
    """cvRound(double) / saturate_cast<int>(double): round-half-even, SSE2 `cvtsd2si` semantics
    (NaN and out-of-range give INT_MIN).  v: float64 array."""
    r = np.rint(v)
    bad = ~((r >= -2147483648.0) & (r <= 2147483647.0))
    out = np.where(bad, -2147483648.0, r)
    return out.astype(np.int64)


def _cv_round_f32(v):
    """cvRound(float): round-half-even on a float32 value, `cvtss2si` semantics."""
    v = np.asarray(v, dtype=np.float32)
    r = np.rint(v).astype(np.float64)
    bad = ~((r >= -2147483648.0) & (r <= 2147483647.0))
    return np.where(bad, -2147483648.0, r).astype(np.int64)


def warp_perspective_rect_mask(rect, H_fwd, frame_width, frame_height):
    """Non-zero pattern of cv2.warpPerspective(mask, H_fwd, (W,H)) for the mask of mfs.py:1050-1052.

    The source mask is 255 on the inclusive rect (L,T,Rt,B) and 0 elsewhere, float64, bilinear,
    BORDER_CONSTANT 0.  Restates imgproc/imgwarp.cpp WarpPerspectiveInvoker + remapBilinear:
    M = invert(H_fwd); destination blocks are 64 wide (OpenCV: min(1024 / min(16, H), W) -- 64 for every frame of at least 16 rows;
    a frame under 16 rows and over 64 pixels wide would differ on exact rounding ties only: not modelled), and for pixel (x, y) of the block starting
    at column xb:  X0 = M0*xb + M1*y + M2 (same for Y0, W0);  W = W0 + M6*x1;  W = W ? 32/W : 0;
    fX = clamp((X0 + M0*x1)*W);  X = cvRound(fX);  ix = X >> 5, fx = X & 31 (same for Y).
    The bilinear sample of the rect image is non-zero iff a tap with non-zero weight lies on the
    rect:  32(L-1) < X < 32(Rt+1)  and  32(T-1) < Y < 32(B+1).
    Returns a bool (H, W) array.
    """
    L, T, Rt, B = rect
    M = invert3x3(np.asarray(H_fwd, dtype=np.float64)).reshape(9)
    xs = np.arange(frame_width, dtype=np.float64)
    xb = np.floor(xs / 64.0) * 64.0
    x1 = xs - xb
    y = np.arange(frame_height, dtype=np.float64)[:, None]
    X0 = (M[0] * xb[None, :] + M[1] * y) + M[2]
    Y0 = (M[3] * xb[None, :] + M[4] * y) + M[5]
    W0 = (M[6] * xb[None, :] + M[7] * y) + M[8]
    Wd = W0 + M[6] * x1[None, :]
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        Ws = np.where(Wd != 0.0, 32.0 / Wd, 0.0)
        fX = np.maximum(float(INT_MIN), np.minimum(float(INT_MAX), (X0 + M[0] * x1[None, :]) * Ws))
        fY = np.maximum(float(INT_MIN), np.minimum(float(INT_MAX), (Y0 + M[3] * x1[None, :]) * Ws))
    X = _cv_round_f64(fX)
    Y = _cv_round_f64(fY)
    return (X > 32 * (L - 1)) & (X < 32 * (Rt + 1)) & (Y > 32 * (T - 1)) & (Y < 32 * (B + 1))


def warp_perspective_f64_bilinear(src, H_fwd, frame_width, frame_height):
    """Full cv2.warpPerspective(src float64 HxW, H_fwd, (W,H)), INTER_LINEAR, BORDER_CONSTANT 0.

    Pixel-value restatement (float weight table as in imgwarp.cpp initInterTab2D); used only by the
    known-answer tests to show that `warp_perspective_rect_mask` is its non-zero pattern.
    """
    src = np.asarray(src, dtype=np.float64)
    sh, sw = src.shape
    M = invert3x3(np.asarray(H_fwd, dtype=np.float64)).reshape(9)
    out = np.zeros((frame_height, frame_width))
    tab = np.array([1.0 - np.float32(i) / np.float32(32) for i in range(32)], dtype=np.float32)
    for yy in range(frame_height):
        for xx in range(frame_width):
            xb = (xx // 64) * 64
            x1 = xx - xb
            X0 = (M[0] * xb + M[1] * yy) + M[2]
            Y0 = (M[3] * xb + M[4] * yy) + M[5]
            W0 = (M[6] * xb + M[7] * yy) + M[8]
            Wd = W0 + M[6] * x1
            Ws = 32.0 / Wd if Wd != 0.0 else 0.0
            fX = max(float(INT_MIN), min(float(INT_MAX), (X0 + M[0] * x1) * Ws))
            fY = max(float(INT_MIN), min(float(INT_MAX), (Y0 + M[3] * x1) * Ws))
            X = int(_cv_round_f64(np.float64(fX)))
            Y = int(_cv_round_f64(np.float64(fY)))
            ix, iy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
            wx = (np.float32(tab[fx]), np.float32(1.0) - np.float32(tab[fx]))
            wy = (np.float32(tab[fy]), np.float32(1.0) - np.float32(tab[fy]))
            if ix >= sw or ix + 1 < 0 or iy >= sh or iy + 1 < 0:
                continue
            acc = 0.0
            for dy in (0, 1):
                for dx in (0, 1):
                    sx, sy = ix + dx, iy + dy
                    v = src[sy, sx] if (0 <= sx < sw and 0 <= sy < sh) else 0.0
                    acc += v * float(np.float32(wy[dy] * wx[dx]))
            out[yy, xx] = acc
    return out


def warp_perspective_f64_bilinear_np(src, H_fwd, frame_width, frame_height):
    """The same arithmetic as `warp_perspective_f64_bilinear`, vectorised (NumPy) so that whole frames are affordable:
    this is what the stub `cv2.warpPerspective` of oracle/gen_golden.py runs inside the REAL reference's per-cell loop
    (mfs.py:1052).  float64 source, float32 weight products from the 32-entry table, taps outside the source = 0,
    sum in the order S00*w0 + S01*w1 + S10*w2 + S11*w3 (imgwarp.cpp remapBilinear, WT = double, AT = float)."""
    src = np.asarray(src, dtype=np.float64)
    sh, sw = src.shape
    M = invert3x3(np.asarray(H_fwd, dtype=np.float64)).reshape(9)
    xs = np.arange(frame_width, dtype=np.float64)
    xb = np.floor(xs / 64.0) * 64.0
    x1 = xs - xb
    y = np.arange(frame_height, dtype=np.float64)[:, None]
    X0 = (M[0] * xb[None, :] + M[1] * y) + M[2]
    Y0 = (M[3] * xb[None, :] + M[4] * y) + M[5]
    W0 = (M[6] * xb[None, :] + M[7] * y) + M[8]
    Wd = W0 + M[6] * x1[None, :]
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        Ws = np.where(Wd != 0.0, 32.0 / Wd, 0.0)
        fX = np.maximum(float(INT_MIN), np.minimum(float(INT_MAX), (X0 + M[0] * x1[None, :]) * Ws))
        fY = np.maximum(float(INT_MIN), np.minimum(float(INT_MAX), (Y0 + M[3] * x1[None, :]) * Ws))
    X = _cv_round_f64(fX)
    Y = _cv_round_f64(fY)
    ix, iy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    tab = np.array([1.0 - np.float32(i) / np.float32(32) for i in range(32)], dtype=np.float32)
    wx = (tab[fx], np.float32(1.0) - tab[fx])
    wy = (tab[fy], np.float32(1.0) - tab[fy])
    acc = np.zeros((frame_height, frame_width))
    for dy in (0, 1):
        for dx in (0, 1):
            tx, ty = ix + dx, iy + dy
            inside = (tx >= 0) & (tx < sw) & (ty >= 0) & (ty < sh)
            v = np.where(inside, src[np.clip(ty, 0, sh - 1), np.clip(tx, 0, sw - 1)], 0.0)
            acc = acc + v * (wy[dy] * wx[dx]).astype(np.float32).astype(np.float64)
    outside = (ix >= sw) | (ix + 1 < 0) | (iy >= sh) | (iy + 1 < 0)
    return np.where(outside, 0.0, acc)


def perspective_transform_f32(points_xy_f32, H):
    """cv2.perspectiveTransform on float32 2-channel points with a float64 3x3 matrix
    (core/matmul.simd.hpp perspectiveTransform_): w = x*m6 + y*m7 + m8 in double; if |w| > FLT_EPSILON
    the outputs are float((x*m0 + y*m1 + m2)*(1/w)), float((x*m3 + y*m4 + m5)*(1/w)), else 0."""
    m = np.asarray(H, dtype=np.float64).reshape(9)
    p = np.asarray(points_xy_f32, dtype=np.float32)
    x = p[..., 0].astype(np.float64)
    y = p[..., 1].astype(np.float64)
    w = (x * m[6] + y * m[7]) + m[8]
    ok = np.abs(w) > FLT_EPSILON
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        iw = 1.0 / w
        u = (((x * m[0] + y * m[1]) + m[2]) * iw).astype(np.float32)
        v = (((x * m[3] + y * m[4]) + m[5]) * iw).astype(np.float32)
    out = np.empty(p.shape, dtype=np.float32)
    out[..., 0] = np.where(ok, u, np.float32(0))
    out[..., 1] = np.where(ok, v, np.float32(0))
    return out


def remap_bilinear_u8c3(src, map_x_f32, map_y_f32, border_bgr):
    """cv2.remap(src uint8 HxWx3, map_x, map_y float32, INTER_LINEAR, BORDER_CONSTANT, borderValue)
    (mfs.py:1063-1069).  Restates imgproc/imgwarp.cpp RemapInvoker + remapBilinear for CV_8UC3:
    sx = cvRound(map_x*32), sy = cvRound(map_y*32); ix = sat_short(sx >> 5), fx = sx & 31 (same y);
    integer weights {(32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx*fy}*32 (sum 2^15); a 2x2 footprint
    wholly outside the image gives the border colour, otherwise each outside tap is replaced by the
    border colour; out = (sum w*S + 2^14) >> 15.  (The table fix-up of initInterTab2D turns the
    fx = fy = 0 entry into {32767,0,0,1}, which cannot change a uint8 result.)"""
    src = np.asarray(src, dtype=np.uint8)
    sh, sw = src.shape[:2]
    mx = np.asarray(map_x_f32, dtype=np.float32)
    my = np.asarray(map_y_f32, dtype=np.float32)
    with np.errstate(over='ignore', invalid='ignore'):
        sx = _cv_round_f32(mx * np.float32(32))
        sy = _cv_round_f32(my * np.float32(32))
    ix = np.clip(sx >> 5, -32768, 32767)
    iy = np.clip(sy >> 5, -32768, 32767)
    fx = sx & 31
    fy = sy & 31
    cval = np.clip(np.rint(np.asarray(border_bgr, dtype=np.float64)), 0, 255).astype(np.int64)
    outside = (ix >= sw) | (ix + 1 < 0) | (iy >= sh) | (iy + 1 < 0)
    acc = np.zeros(mx.shape + (3,), dtype=np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            tx = ix + dx
            ty = iy + dy
            inside = (tx >= 0) & (tx < sw) & (ty >= 0) & (ty < sh)
            tap = src[np.clip(ty, 0, sh - 1), np.clip(tx, 0, sw - 1)].astype(np.int64)
            tap = np.where(inside[..., None], tap, cval)
            wx = fx if dx else 32 - fx
            wy = fy if dy else 32 - fy
            acc += (wx * wy * 32)[..., None] * tap
    out = (acc + (1 << 14)) >> 15
    out = np.where(outside[..., None], cval, out)
    return out.astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# Warp half: the reference's per-cell painter loop
# ------------------------------------------------------------------------------------------------

def cell_tables(frame_width, frame_height, mesh_rows, mesh_cols, unstab_disp_f, stab_disp_f):
    """Per-cell (H_fwd, H_inv, rect) of one frame: mfs.py:948-967, 1025-1027, 1039-1048."""
    grid = vertex_x_y(frame_width, frame_height, mesh_rows, mesh_cols)              # mfs.py:948
    rc_unstab = np.reshape(grid, (mesh_rows + 1, mesh_cols + 1, 2))                 # mfs.py:953
    motion = np.reshape(np.asarray(stab_disp_f, dtype=np.float64) - np.asarray(unstab_disp_f, dtype=np.float64),
                        (-1, 1, 2))                                                 # mfs.py:964-967
    stab = grid + motion                                                            # mfs.py:1025 (float64)
    rc_stab = np.reshape(stab, (mesh_rows + 1, mesh_cols + 1, 2))                   # mfs.py:1027
    cells = []
    for r in range(mesh_rows):                                                      # mfs.py:1031
        for c in range(mesh_cols):                                                  # mfs.py:1032
            ub = rc_unstab[r:r + 2, c:c + 2].reshape(-1, 2)                         # mfs.py:1039
            sb = rc_stab[r:r + 2, c:c + 2].reshape(-1, 2)                           # mfs.py:1040
            Hf = find_homography_4pt(ub, sb)                                        # mfs.py:1041
            Hi = find_homography_4pt(sb, ub)                                        # mfs.py:1042
            xs, ys = np.transpose(ub)                                               # mfs.py:1044
            rect = (math.floor(np.min(xs)), math.floor(np.min(ys)),
                    math.ceil(np.max(xs)), math.ceil(np.max(ys)))                   # mfs.py:1045-1048 (L,T,Rt,B)
            cells.append((Hf, Hi, rect))
    return cells


def warp_frame(frame, mesh_rows, mesh_cols, unstab_disp_f, stab_disp_f, border_bgr=(0, 0, 255), max_cells=None,
               full_mask=False):
    """One iteration of the frame loop mfs.py:1000-1100.

    Returns (stabilized_frame uint8 HxWx3, (left, top, right, bottom) per-frame crop values,
    map_x float64 HxW, map_y float64 HxW).
    max_cells (TIMING ONLY, bench.py's reference-faithful CPU leg): stop the painter loop after that many cells -- every
    cell costs the same full-frame passes, so the loop time scales linearly; the frame returned is then incomplete.
    full_mask: build the cell mask like the reference does (full float64 mask image through the bilinear warp,
    mfs.py:1050-1052) instead of through its non-zero pattern; same result (tests/test_oracle_warp_kat.py)."""
    frame = np.asarray(frame, dtype=np.uint8)
    H, W = frame.shape[:2]
    map_x = np.full((H, W), W + 1)                                                  # mfs.py:983, 1017
    map_y = np.full((H, W), H + 1)                                                  # mfs.py:984, 1018
    xy = np.swapaxes(np.indices((W, H), dtype=np.float32), 0, 2)                    # mfs.py:985  [y][x] = (x, y)
    for k, (Hf, Hi, (L, T, Rt, B)) in enumerate(cell_tables(W, H, mesh_rows, mesh_cols, unstab_disp_f, stab_disp_f)):
        if max_cells is not None and k >= max_cells:
            break
        if Hf is None or Hi is None:
            raise ValueError('degenerate mesh cell (cv2.findHomography would return None)')
        rect = (max(L, 0), max(T, 0), min(Rt, W - 1), min(B, H - 1))                # numpy slice clipping, mfs.py:1051
        if full_mask:
            cell_mask = np.zeros((H, W))                                            # mfs.py:1050
            cell_mask[rect[1]:rect[3] + 1, rect[0]:rect[2] + 1] = 255               # mfs.py:1051
            mask = warp_perspective_f64_bilinear_np(cell_mask, Hf, W, H)            # mfs.py:1052
        else:
            mask = warp_perspective_rect_mask(rect, Hf, W, H)                       # mfs.py:1050-1052
        cell_xy = perspective_transform_f32(xy, Hi)                                 # mfs.py:1054-1056
        map_x = np.where(mask, cell_xy[..., 0], map_x)                              # mfs.py:1060 (-> float64)
        map_y = np.where(mask, cell_xy[..., 1], map_y)                              # mfs.py:1061
    map_x = map_x.astype(np.float64)
    map_y = map_y.astype(np.float64)
    out = remap_bilinear_u8c3(frame, map_x.astype(np.float32), map_y.astype(np.float32), border_bgr)  # mfs.py:1063-1069
    left, right, top, bottom = 0, W - 1, 0, H - 1                                   # mfs.py:992-995
    c = np.where(np.abs(map_x - 0) < 1)[1]                                          # mfs.py:1075
    if c.size > 0:
        left = int(np.max(c))
    c = np.where(np.abs(map_x - (W - 1)) < 1)[1]                                    # mfs.py:1082
    if c.size > 0:
        right = int(np.min(c))
    c = np.where(np.abs(map_y - 0) < 1)[0]                                          # mfs.py:1089
    if c.size > 0:
        top = int(np.max(c))
    c = np.where(np.abs(map_y - (H - 1)) < 1)[0]                                    # mfs.py:1096
    if c.size > 0:
        bottom = int(np.min(c))
    return out, (left, top, right, bottom), map_x, map_y


def stabilized_frames_and_crop_boundaries(frames, mesh_rows, mesh_cols, unstab_disp, stab_disp,
                                          border_bgr=(0, 0, 255)):
    """mfs.py:909-1108.  Returns (list of frames, (left, top, right, bottom), per-frame crops (F,4))."""
    outs = []
    per_frame = []
    for f, frame in enumerate(frames):
        out, crop, _, _ = warp_frame(frame, mesh_rows, mesh_cols, unstab_disp[f], stab_disp[f], border_bgr)
        outs.append(out)
        per_frame.append(crop)
    pf = np.asarray(per_frame, dtype=np.int64).reshape(-1, 4)
    bounds = (int(pf[:, 0].max()), int(pf[:, 1].max()), int(pf[:, 2].min()), int(pf[:, 3].min()))  # mfs.py:1103-1106
    return outs, bounds, pf


# ------------------------------------------------------------------------------------------------
# Next row: crop + resize  (mfs.py:1111-1157)
# ------------------------------------------------------------------------------------------------

def resize_linear_tables(src_len, dst_len):
    """Per-output-index source offset and 11-bit fixed-point weights of cv2.resize INTER_LINEAR for 8-bit
    images (imgproc/resize.cpp, cv::hal::resize): scale = 1/((double)dst/src); f = float((d+0.5)*scale-0.5);
    s = floor(f); f -= s; s < 0 -> (s, f) = (0, 0); s >= src-1 -> (src-1, 0) [x axis only; the y axis clips
    its two row indices instead]; weights = cvRound(float(1-f)*2048), cvRound(f*2048) as int16."""
    inv_scale = float(dst_len) / float(src_len)
    scale = 1.0 / inv_scale
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f):
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return w0, w1


def resize_linear_u8(src, dst_w, dst_h):
    """cv2.resize(src uint8 HxWxC, (dst_w, dst_h)) with the default INTER_LINEAR (mfs.py:1150-1155; the fx/fy
    arguments are ignored because dsize is given).  Two-pass fixed point exactly as resize.cpp:
    horizontal t = S[sx]*a0 + S[sx+1]*a1 (int32, scale 2^11); vertical
    out = (((b0*(t0 >> 4)) >> 16) + ((b1*(t1 >> 4)) >> 16) + 2) >> 2."""
    src = np.asarray(src, dtype=np.uint8)
    sh, sw = src.shape[:2]
    if sh == 0 or sw == 0:
        raise ValueError('cv2.resize: empty source (the crop rectangle is empty)')
    sx, fx = resize_linear_tables(sw, dst_w)
    low = sx < 0
    sx = np.where(low, 0, sx); fx = np.where(low, np.float32(0), fx)
    high = sx >= sw - 1
    sx = np.where(high, sw - 1, sx); fx = np.where(high, np.float32(0), fx)
    a0, a1 = _coef(fx.astype(np.float32))
    sy, fy = resize_linear_tables(sh, dst_h)
    b0, b1 = _coef(fy)
    sy0 = np.clip(sy, 0, sh - 1)
    sy1 = np.clip(sy + 1, 0, sh - 1)
    sx1 = np.minimum(sx + 1, sw - 1)                 # where sx = sw-1 the second weight is 0
    S = src.astype(np.int64)
    t0 = S[sy0][:, sx] * a0[None, :, None] + S[sy0][:, sx1] * a1[None, :, None]
    t1 = S[sy1][:, sx] * a0[None, :, None] + S[sy1][:, sx1] * a1[None, :, None]
    out = (((b0[:, None, None] * (t0 >> 4)) >> 16) + ((b1[:, None, None] * (t1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_frames(uncropped_frames, crop_boundaries):
    """mfs.py:1111-1157: crop every frame to the inclusive bounds and resize back to (W, H)."""
    frame_height, frame_width = uncropped_frames[0].shape[:2]
    left, top, right, bottom = (int(v) for v in crop_boundaries)
    return [resize_linear_u8(f[top:bottom + 1, left:right + 1], frame_width, frame_height) for f in uncropped_frames]

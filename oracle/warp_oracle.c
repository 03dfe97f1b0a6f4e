/* CPU oracle (plain C, float64 / integer) for the MeshFlow hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Single-pass restatement of the same arithmetic as oracle/meshflow_oracle.py (which follows the
 * reference's per-cell painter loop literally and is checked against this file in
 * tests/test_oracle_c.py).  `mfs.py:N` = /root/reference/meshflowstabilizer.py line N.
 *
 *   mfo_jacobi_banded      mfs.py:844-878 (banded form of x <- diag(1/on) (b - off x))
 *   mfo_cell_table         mfs.py:881-906, 964-967, 1025-1027, 1039-1048 + cv2.findHomography (4 points)
 *   mfo_warp_frame         mfs.py:1017-1019, 1050-1069 (warpPerspective mask, perspectiveTransform,
 *                          painter merge, remap) and mfs.py:1075-1098 (crop-boundary scan)
 *
 * Pinning: the Jacobi routine is checked against vectors produced by the reference itself
 * (tests/golden/); the warp routines restate OpenCV semantics from memory of its sources and are
 * "parity unpinned" (no cv2 in this image) -- see the header of meshflow_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA contraction so that every routine is
 * bit-reproducible and comparable with the HIP kernels, which are built the same way; the one place
 * that uses fused multiply-add, the Jacobi tap sum, calls fma() explicitly on both sides).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define MFO_CELL_DOUBLES 32   /* record: M[9] Hi[9] rect(L,T,Rt,B) bbox(x0,y0,x1,y1) status pad[5] */
#define MFO_OFF_M 0
#define MFO_OFF_HI 9
#define MFO_OFF_RECT 18
#define MFO_OFF_BBOX 22
#define MFO_OFF_STATUS 26

/* ---------------------------------------------------------------------------------------------- */
/* Jacobi (mfs.py:871-878).  b, x: [F][S] frame-major.  x_new[t] = inv_on[t]*(b[t] + 2*lam[t]*s),   */
/* s = sum_{d=-omega..omega, 0<=t+d<F} taps[d+omega]*x[t+d], accumulated in increasing d with fma.  */
/* (off[t,t+d] = -2*lam[t]*taps[d], band includes d = 0, mfs.py:767-781.)                           */
/* ---------------------------------------------------------------------------------------------- */
void mfo_jacobi_banded(const double* b, double* x_out, const double* taps, const double* lam,
                       const double* inv_on, int F, int S, int omega, int iters)
{
    /* The S series are independent (one vertex component each, mfs.py:695-704): blocks of series are swept
       through all iterations by one thread each, with no synchronisation; per series the arithmetic and its
       order are exactly as in the single-threaded loop. */
    const int BS = 8;
    const int nblocks = (S + BS - 1) / BS;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
    for (int blk = 0; blk < nblocks; ++blk) {
        const int s0 = blk * BS, ns = (s0 + BS <= S) ? BS : S - s0;
        double* cur = (double*)malloc((size_t)F * BS * sizeof(double));
        double* nxt = (double*)malloc((size_t)F * BS * sizeof(double));
        for (int t = 0; t < F; ++t)
            for (int s = 0; s < ns; ++s) cur[(size_t)t * BS + s] = b[(size_t)t * S + s0 + s];   /* x_start = b, mfs.py:699-703, 871 */
        for (int it = 0; it < iters; ++it) {
            for (int t = 0; t < F; ++t) {
                const double two_lam = 2.0 * lam[t];
                for (int s = 0; s < ns; ++s) {
                    double acc = 0.0;
                    for (int d = -omega; d <= omega; ++d) {
                        const int r = t + d;
                        if (r < 0 || r >= F) continue;
                        acc = fma(taps[d + omega], cur[(size_t)r * BS + s], acc);
                    }
                    nxt[(size_t)t * BS + s] = inv_on[t] * fma(two_lam, acc, b[(size_t)t * S + s0 + s]);
                }
            }
            double* tmp = cur; cur = nxt; nxt = tmp;
        }
        for (int t = 0; t < F; ++t)
            for (int s = 0; s < ns; ++s) x_out[(size_t)t * S + s0 + s] = cur[(size_t)t * BS + s];
        free(cur); free(nxt);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* cv2.findHomography for 4 points (method 0): normalised DLT, see find_homography_4pt in           */
/* meshflow_oracle.py for the derivation.  Points are first rounded to float32.                     */
/* ---------------------------------------------------------------------------------------------- */
/* Unit square (0,0), (1,0), (1,1), (0,1) -> p0, p1, p2, p3 (Heckbert 1989): row-major {a, b, c, d, e, f, g, h, 1}; 0 when p1, p2, p3
 * are collinear.  The operation order of _square_to_quad in meshflow_oracle.py. */
static int square_to_quad(const double p0[2], const double p1[2], const double p2[2], const double p3[2], double S[9])
{
    const double sx = ((p0[0] - p1[0]) + p2[0]) - p3[0];
    const double sy = ((p0[1] - p1[1]) + p2[1]) - p3[1];
    const double dx1 = p1[0] - p2[0], dx2 = p3[0] - p2[0];
    const double dy1 = p1[1] - p2[1], dy2 = p3[1] - p2[1];
    const double den = dx1 * dy2 - dx2 * dy1;
    if (!(fabs(den) > 1e-10)) return 0;          /* p1, p2, p3 collinear (normalised coordinates: O(1)) */
    const double g = (sx * dy2 - dx2 * sy) / den;
    const double h = (dx1 * sy - sx * dy1) / den;
    S[0] = (p1[0] - p0[0]) + g * p1[0]; S[1] = (p3[0] - p0[0]) + h * p3[0]; S[2] = p0[0];
    S[3] = (p1[1] - p0[1]) + g * p1[1]; S[4] = (p3[1] - p0[1]) + h * p3[1]; S[5] = p0[1];
    S[6] = g; S[7] = h; S[8] = 1.0;
    return 1;
}

/* adjugate of a row-major 3x3 whose last entry is 1 (_adjugate3) */
static void adjugate3(const double S[9], double A[9])
{
    const double a = S[0], b = S[1], c = S[2], d = S[3], e = S[4], f = S[5], g = S[6], h = S[7];
    A[0] = e - f * h; A[1] = c * h - b; A[2] = b * f - c * e;
    A[3] = f * g - d; A[4] = a - c * g; A[5] = c * d - a * f;
    A[6] = d * h - e * g; A[7] = b * g - a * h; A[8] = a * e - b * d;
}

static void matmul3(const double a[9], const double b[9], double c[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = a[i * 3 + 0] * b[0 * 3 + j];
            s = s + a[i * 3 + 1] * b[1 * 3 + j];
            s = s + a[i * 3 + 2] * b[2 * 3 + j];
            c[i * 3 + j] = s;
        }
}

int mfo_find_homography_4pt(const double src[8], const double dst[8], double H[9])
{
    double Mx[4], My[4], mx[4], my[4];
    for (int i = 0; i < 4; ++i) {
        Mx[i] = (double)(float)src[2 * i]; My[i] = (double)(float)src[2 * i + 1];
        mx[i] = (double)(float)dst[2 * i]; my[i] = (double)(float)dst[2 * i + 1];
    }
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
    for (int i = 0; i < 4; ++i) { cmx += mx[i]; cmy += my[i]; cMx += Mx[i]; cMy += My[i]; }
    cmx /= 4; cmy /= 4; cMx /= 4; cMy /= 4;
    double smx = 0, smy = 0, sMx = 0, sMy = 0;
    for (int i = 0; i < 4; ++i) {
        smx += fabs(mx[i] - cmx); smy += fabs(my[i] - cmy);
        sMx += fabs(Mx[i] - cMx); sMy += fabs(My[i] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = 4 / smx; smy = 4 / smy; sMx = 4 / sMx; sMy = 4 / sMy;
    double invHnorm[9] = { 1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1 };
    double Hnorm2[9] = { sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1 };
    /* the 4-point problem in normalised coordinates, in closed form (find_homography_4pt, solver 'closed'): corners arrive as TL, TR,
     * BL, BR; the unit square's cyclic order is points 0, 1, 3, 2 */
    double nm[4][2], nM[4][2];
    for (int i = 0; i < 4; ++i) {
        nm[i][0] = (mx[i] - cmx) * smx; nm[i][1] = (my[i] - cmy) * smy;
        nM[i][0] = (Mx[i] - cMx) * sMx; nM[i][1] = (My[i] - cMy) * sMy;
    }
    double Sd[9], Ss[9], As[9], H0[9];
    if (!square_to_quad(nm[0], nm[1], nm[3], nm[2], Sd) || !square_to_quad(nM[0], nM[1], nM[3], nM[2], Ss)) return 0;
    adjugate3(Ss, As);
    {   /* either quad with ANY three corners collinear: its square -> quad map is singular (normalised coordinates are O(1)) */
        const double det_s = (Ss[0] * As[0] + Ss[1] * As[3]) + Ss[2] * As[6];
        const double det_d = (Sd[0] * (Sd[4] - Sd[5] * Sd[7]) + Sd[1] * (Sd[5] * Sd[6] - Sd[3])) + Sd[2] * (Sd[3] * Sd[7] - Sd[4] * Sd[6]);
        if (!(fabs(det_s) > 1e-10) || !(fabs(det_d) > 1e-10)) return 0;
    }
    matmul3(Sd, As, H0);
    double Ht[9], Hd[9];
    matmul3(invHnorm, H0, Ht);
    matmul3(Ht, Hnorm2, Hd);
    if (Hd[8] == 0.0) return 0;
    double sc = 1.0 / Hd[8];
    for (int i = 0; i < 9; ++i) H[i] = Hd[i] * sc;
    return 1;
}

/* cv::invert, 3x3 CV_64F closed form (core/lapack.cpp). */
void mfo_invert3x3(const double S[9], double t[9])
{
#define Sd(y, x) S[(y) * 3 + (x)]
    double d = Sd(0,0) * (Sd(1,1) * Sd(2,2) - Sd(1,2) * Sd(2,1))
             - Sd(0,1) * (Sd(1,0) * Sd(2,2) - Sd(1,2) * Sd(2,0))
             + Sd(0,2) * (Sd(1,0) * Sd(2,1) - Sd(1,1) * Sd(2,0));
    if (d == 0.0) { for (int i = 0; i < 9; ++i) t[i] = 0; return; }
    d = 1.0 / d;
    t[0] = (Sd(1,1) * Sd(2,2) - Sd(1,2) * Sd(2,1)) * d;
    t[1] = (Sd(0,2) * Sd(2,1) - Sd(0,1) * Sd(2,2)) * d;
    t[2] = (Sd(0,1) * Sd(1,2) - Sd(0,2) * Sd(1,1)) * d;
    t[3] = (Sd(1,2) * Sd(2,0) - Sd(1,0) * Sd(2,2)) * d;
    t[4] = (Sd(0,0) * Sd(2,2) - Sd(0,2) * Sd(2,0)) * d;
    t[5] = (Sd(0,2) * Sd(1,0) - Sd(0,0) * Sd(1,2)) * d;
    t[6] = (Sd(1,0) * Sd(2,1) - Sd(1,1) * Sd(2,0)) * d;
    t[7] = (Sd(0,1) * Sd(2,0) - Sd(0,0) * Sd(2,1)) * d;
    t[8] = (Sd(0,0) * Sd(1,1) - Sd(0,1) * Sd(1,0)) * d;
#undef Sd
}

/* Vertex grid, mfs.py:901-906: ceil((W-1)*(col/C)), ceil((H-1)*(row/R)), float32 (integers). */
static double grid_x(int W, int C, int col) { return ceil((double)(W - 1) * ((double)col / (double)C)); }
static double grid_y(int H, int R, int row) { return ceil((double)(H - 1) * ((double)row / (double)R)); }

/* Conservative bounding box (inclusive, clamped to the frame) of the pixels where the cell's warped
 * mask can be non-zero: the forward image of the rect dilated by one pixel, +-2 px.  Falls back to the
 * whole frame when the projective map is not well behaved on the rect or on the frame.  The brute-force
 * mode of mfo_warp_frame does not use it; the HIP kernel's candidate culling does, and computes it the
 * same way, so the two tables can be compared entry by entry. */
static void cell_bbox(const double Hf[9], const double M[9], double L, double T, double Rt, double B,
                      int W, int H, double bbox[4])
{
    double qx[4] = { L - 1, Rt + 1, L - 1, Rt + 1 };
    double qy[4] = { T - 1, T - 1, B + 1, B + 1 };
    double fx[4] = { 0, (double)(W - 1), 0, (double)(W - 1) };
    double fy[4] = { 0, 0, (double)(H - 1), (double)(H - 1) };
    int ok = 1;
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    for (int i = 0; i < 4; ++i) {
        double den = (Hf[6] * qx[i] + Hf[7] * qy[i]) + Hf[8];
        double wf = (M[6] * fx[i] + M[7] * fy[i]) + M[8];
        if (!(den > 1e-3) || !(wf > 1e-3)) { ok = 0; break; }
        double px = ((Hf[0] * qx[i] + Hf[1] * qy[i]) + Hf[2]) / den;
        double py = ((Hf[3] * qx[i] + Hf[4] * qy[i]) + Hf[5]) / den;
        if (!(fabs(px) < 1e9) || !(fabs(py) < 1e9)) { ok = 0; break; }
        if (i == 0) { x0 = x1 = px; y0 = y1 = py; }
        else {
            if (px < x0) x0 = px;
            if (px > x1) x1 = px;
            if (py < y0) y0 = py;
            if (py > y1) y1 = py;
        }
    }
    if (!ok) { bbox[0] = 0; bbox[1] = 0; bbox[2] = W - 1; bbox[3] = H - 1; return; }
    x0 = floor(x0) - 2; y0 = floor(y0) - 2; x1 = ceil(x1) + 2; y1 = ceil(y1) + 2;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > W - 1) x1 = W - 1;
    if (y1 > H - 1) y1 = H - 1;
    if (x0 > x1 || y0 > y1) { bbox[0] = 1; bbox[1] = 1; bbox[2] = 0; bbox[3] = 0; return; }   /* empty */
    bbox[0] = x0; bbox[1] = y0; bbox[2] = x1; bbox[3] = y1;
}

/* Per-cell table of one frame.  unstab/stab: [(R+1)*(C+1)][2] float64 displacements of that frame.
 * Returns the number of degenerate cells (cv2.findHomography would return None for them). */
int mfo_cell_table(int W, int H, int R, int C, const double* unstab, const double* stab, double* table)
{
    int bad = 0;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            double* rec = table + (size_t)(r * C + c) * MFO_CELL_DOUBLES;
            double ub[8], sb[8];
            for (int k = 0; k < 4; ++k) {                       /* TL, TR, BL, BR: mfs.py:1039-1040 */
                int rr = r + (k >> 1), cc = c + (k & 1);
                int v = rr * (C + 1) + cc;
                double gx = (double)(float)grid_x(W, C, cc), gy = (double)(float)grid_y(H, R, rr);
                ub[2 * k] = gx; ub[2 * k + 1] = gy;
                sb[2 * k] = gx + (stab[2 * v] - unstab[2 * v]);          /* mfs.py:964-967, 1025 */
                sb[2 * k + 1] = gy + (stab[2 * v + 1] - unstab[2 * v + 1]);
            }
            double Hf[9], Hi[9];
            int ok = mfo_find_homography_4pt(ub, sb, Hf) && mfo_find_homography_4pt(sb, ub, Hi);
            memset(rec, 0, MFO_CELL_DOUBLES * sizeof(double));
            double L = floor(fmin(fmin(ub[0], ub[2]), fmin(ub[4], ub[6])));     /* mfs.py:1045-1048 */
            double Rt = ceil(fmax(fmax(ub[0], ub[2]), fmax(ub[4], ub[6])));
            double T = floor(fmin(fmin(ub[1], ub[3]), fmin(ub[5], ub[7])));
            double B = ceil(fmax(fmax(ub[1], ub[3]), fmax(ub[5], ub[7])));
            rec[MFO_OFF_RECT + 0] = L; rec[MFO_OFF_RECT + 1] = T; rec[MFO_OFF_RECT + 2] = Rt; rec[MFO_OFF_RECT + 3] = B;
            if (!ok) {
                rec[MFO_OFF_STATUS] = 1; ++bad;
                rec[MFO_OFF_BBOX + 0] = 1; rec[MFO_OFF_BBOX + 1] = 1; rec[MFO_OFF_BBOX + 2] = 0; rec[MFO_OFF_BBOX + 3] = 0;
                continue;
            }
            mfo_invert3x3(Hf, rec + MFO_OFF_M);                 /* warpPerspective inverts H_fwd */
            memcpy(rec + MFO_OFF_HI, Hi, 9 * sizeof(double));
            cell_bbox(Hf, rec + MFO_OFF_M, L, T, Rt, B, W, H, rec + MFO_OFF_BBOX);
        }
    return bad;
}

/* cvRound semantics of the SSE2 builds: round half to even, NaN / out of range -> INT_MIN. */
static int32_t cv_round_f64(double v)
{
    double r = nearbyint(v);
    if (!(r >= -2147483648.0 && r <= 2147483647.0)) return INT32_MIN;
    return (int32_t)r;
}
static int32_t cv_round_f32(float v)
{
    float r = nearbyintf(v);
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return INT32_MIN;
    return (int32_t)r;
}
static int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* Mask test of one cell at destination pixel (x, y): warpPerspective's coordinate arithmetic
 * (64-wide blocks) followed by "does the bilinear footprint touch the rect with non-zero weight". */
static int mask_test(const double* rec, int x, int y)
{
    const double* M = rec + MFO_OFF_M;
    double xb = (double)(x & ~63), x1 = (double)(x & 63), yy = (double)y;
    double X0 = (M[0] * xb + M[1] * yy) + M[2];
    double Y0 = (M[3] * xb + M[4] * yy) + M[5];
    double W0 = (M[6] * xb + M[7] * yy) + M[8];
    double Wd = W0 + M[6] * x1;
    double Ws = Wd != 0.0 ? 32.0 / Wd : 0.0;
    double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * Ws));
    double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * Ws));
    /* std::min/std::max as OpenCV writes them send a NaN product to INT_MAX; fmin/fmax do the same. */
    int64_t X = cv_round_f64(fX);
    int64_t Y = cv_round_f64(fY);
    int64_t L = (int64_t)rec[MFO_OFF_RECT + 0], T = (int64_t)rec[MFO_OFF_RECT + 1];
    int64_t Rt = (int64_t)rec[MFO_OFF_RECT + 2], B = (int64_t)rec[MFO_OFF_RECT + 3];
    return X > 32 * (L - 1) && X < 32 * (Rt + 1) && Y > 32 * (T - 1) && Y < 32 * (B + 1);
}

/* One frame: mfs.py:1017-1019 + 1031-1098 as a single pass over destination pixels.
 * owner(pixel) = last cell in row-major order whose mask test passes (painter merge, mfs.py:1060-1061).
 * use_bbox = 0: every cell is tested for every pixel (descending, first hit wins) -- the checker.
 * use_bbox = 1: cells whose recorded bounding box misses the pixel are skipped -- the faster variant
 * timed as the CPU baseline; tests assert that both give identical output.
 * crop: {left, top, right, bottom} of this frame (defaults 0, 0, W-1, H-1; mfs.py:992-995).
 * map_x/map_y (optional): the float32 maps handed to remap. */
void mfo_warp_frame(const uint8_t* src, uint8_t* dst, int W, int H, int R, int C, const double* table,
                    const uint8_t border[3], int use_bbox, int32_t crop[4], float* map_x, float* map_y)
{
    int ncell = R * C;
    int left = 0, top = 0, right = W - 1, bottom = H - 1;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float u = (float)(W + 1), v = (float)(H + 1);                   /* mfs.py:983-984 */
            for (int k = ncell - 1; k >= 0; --k) {
                const double* rec = table + (size_t)k * MFO_CELL_DOUBLES;
                if (rec[MFO_OFF_STATUS] != 0.0) continue;
                if (use_bbox && (x < rec[MFO_OFF_BBOX] || x > rec[MFO_OFF_BBOX + 2] ||
                                 y < rec[MFO_OFF_BBOX + 1] || y > rec[MFO_OFF_BBOX + 3]))
                    continue;
                if (!mask_test(rec, x, y)) continue;
                const double* m = rec + MFO_OFF_HI;                          /* perspectiveTransform, mfs.py:1054 */
                double xs = (double)(float)x, ys = (double)(float)y;
                double w = (xs * m[6] + ys * m[7]) + m[8];
                if (fabs(w) > (double)FLT_EPSILON) {
                    w = 1. / w;
                    u = (float)(((xs * m[0] + ys * m[1]) + m[2]) * w);
                    v = (float)(((xs * m[3] + ys * m[4]) + m[5]) * w);
                } else {
                    u = 0.f; v = 0.f;
                }
                break;
            }
            if (map_x) map_x[(size_t)y * W + x] = u;
            if (map_y) map_y[(size_t)y * W + x] = v;
            /* crop-boundary scan on the float64 maps, mfs.py:1075-1098 */
            if (fabs((double)u - 0.0) < 1.0 && x > left) left = x;
            if (fabs((double)u - (double)(W - 1)) < 1.0 && x < right) right = x;
            if (fabs((double)v - 0.0) < 1.0 && y > top) top = y;
            if (fabs((double)v - (double)(H - 1)) < 1.0 && y < bottom) bottom = y;
            /* remap, INTER_LINEAR, BORDER_CONSTANT: mfs.py:1063-1069 */
            int32_t sx = cv_round_f32(u * 32.0f), sy = cv_round_f32(v * 32.0f);
            int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5);
            int fx = sx & 31, fy = sy & 31;
            uint8_t* o = dst + ((size_t)y * W + x) * 3;
            if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) {
                o[0] = border[0]; o[1] = border[1]; o[2] = border[2];
                continue;
            }
            int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
            int in_x0 = ix >= 0 && ix < W, in_x1 = ix + 1 >= 0 && ix + 1 < W;
            int in_y0 = iy >= 0 && iy < H, in_y1 = iy + 1 >= 0 && iy + 1 < H;
            const uint8_t* p00 = (in_x0 && in_y0) ? src + ((size_t)iy * W + ix) * 3 : border;
            const uint8_t* p01 = (in_x1 && in_y0) ? src + ((size_t)iy * W + ix + 1) * 3 : border;
            const uint8_t* p10 = (in_x0 && in_y1) ? src + ((size_t)(iy + 1) * W + ix) * 3 : border;
            const uint8_t* p11 = (in_x1 && in_y1) ? src + ((size_t)(iy + 1) * W + ix + 1) * 3 : border;
            for (int ch = 0; ch < 3; ++ch) {
                int acc = (w00 * p00[ch] + w01 * p01[ch] + w10 * p10[ch] + w11 * p11[ch]) * 32;
                o[ch] = (uint8_t)((acc + (1 << 14)) >> 15);
            }
        }
    /* the "(left == default) means empty" distinction does not matter: max(0, ...) = default */
    crop[0] = left; crop[1] = top; crop[2] = right; crop[3] = bottom;
}

/* Whole clip, optionally multi-threaded over frames (OpenMP when compiled with -fopenmp).
 * frames/out: [n][H][W][3]; unstab/stab: [n][V][2]; crop: [n][4].  Returns degenerate-cell count. */
int mfo_warp_clip(const uint8_t* frames, uint8_t* out, int n, int W, int H, int R, int C,
                  const double* unstab, const double* stab, const uint8_t border[3], int use_bbox,
                  int32_t* crop)
{
    int bad = 0;
    size_t fsz = (size_t)W * H * 3, vsz = (size_t)(R + 1) * (C + 1) * 2;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) reduction(+:bad)
#endif
    for (int f = 0; f < n; ++f) {
        double* table = (double*)malloc((size_t)R * C * MFO_CELL_DOUBLES * sizeof(double));
        bad += mfo_cell_table(W, H, R, C, unstab + f * vsz, stab + f * vsz, table);
        mfo_warp_frame(frames + f * fsz, out + f * fsz, W, H, R, C, table, border, use_bbox, crop + 4 * f, 0, 0);
        free(table);
    }
    return bad;
}

int mfo_cell_doubles(void) { return MFO_CELL_DOUBLES; }

/* Thread count of the OpenMP build (the environment may pin OMP_NUM_THREADS, e.g. torchrun sets it to 1). */
#ifdef _OPENMP
#include <omp.h>
int mfo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int mfo_set_threads(int n) { (void)n; return 1; }
#endif

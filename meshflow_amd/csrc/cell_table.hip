// Kernel 2a: per-cell homography table.  One thread per (frame, cell).
//
// Reference (meshflowstabilizer.py): vertex grid :881-906; stabilized vertex positions :964-967, :1025;
// two independent cv2.findHomography calls per cell :1041-1042; cell rect :1045-1048; and the matrix
// inversion cv2.warpPerspective applies to the forward homography (:1052).
//
// cv2.findHomography with 4 points is the normalised DLT (OpenCV calib3d/fundam.cpp runKernel): inputs
// rounded to float32, zero-centroid / unit-mean-absolute-deviation normalisation, 8 equations in 9
// unknowns, de-normalisation invHnorm * H0 * Hnorm2 and scaling by 1/H[2][2].  With exactly four points the 8 equations have one
// solution; it is taken here in closed form -- square -> quad of the normalised destination points times the adjugate of square ->
// quad of the normalised source points (Heckbert) -- in a fixed operation order shared with the CPU oracle, float64, no FMA
// contraction (the file is compiled with -ffp-contract=off), so the table is bit-identical to the oracle's.  (Rounds 1-3 solved the
// 8 x 8 system by Gaussian elimination with predicated row exchanges: ~3,500 dependent instructions per homography against ~500.)
//
// The work is tiny (2*R*C 8x8 solves per frame); the kernel exists so that the Jacobi output never leaves
// the device between the two halves of the path.
#include "mf_common.h"

#ifndef MF_PLAN_EXP
#define MF_PLAN_EXP 0          // 1-3: timing-only builds of the plan kernel (tools/plan_profile.sh); nothing of them is in the product
#endif

namespace mf {

// Unit square (0,0), (1,0), (1,1), (0,1) -> p0, p1, p2, p3 (Heckbert 1989: the 8 equations of the 4-point problem solved by hand):
// row-major {a, b, c, d, e, f, g, h, 1}; false when p1, p2, p3 are collinear.  Operation order = oracle/warp_oracle.c square_to_quad.
__device__ static bool square_to_quad(const double p0[2], const double p1[2], const double p2[2], const double p3[2], double S[9])
{
    const double sx = ((p0[0] - p1[0]) + p2[0]) - p3[0];
    const double sy = ((p0[1] - p1[1]) + p2[1]) - p3[1];
    const double dx1 = p1[0] - p2[0], dx2 = p3[0] - p2[0];
    const double dy1 = p1[1] - p2[1], dy2 = p3[1] - p2[1];
    const double den = dx1 * dy2 - dx2 * dy1;
    const bool ok = fabs(den) > 1e-10;            // (p1, p2, p3 collinear; the caller normalises the points: coordinates of O(1))
    const double g = (sx * dy2 - dx2 * sy) / den;
    const double h = (dx1 * sy - sx * dy1) / den;
    S[0] = (p1[0] - p0[0]) + g * p1[0]; S[1] = (p3[0] - p0[0]) + h * p3[0]; S[2] = p0[0];
    S[3] = (p1[1] - p0[1]) + g * p1[1]; S[4] = (p3[1] - p0[1]) + h * p3[1]; S[5] = p0[1];
    S[6] = g; S[7] = h; S[8] = 1.0;
    return ok;
}

// adjugate of a row-major 3 x 3 whose last entry is 1
__device__ static void adjugate3(const double S[9], double A[9])
{
    const double a = S[0], b = S[1], c = S[2], d = S[3], e = S[4], f = S[5], g = S[6], h = S[7];
    A[0] = e - f * h; A[1] = c * h - b; A[2] = b * f - c * e;
    A[3] = f * g - d; A[4] = a - c * g; A[5] = c * d - a * f;
    A[6] = d * h - e * g; A[7] = b * g - a * h; A[8] = a * e - b * d;
}

__device__ static void matmul3(const double a[9], const double b[9], double c[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = a[i * 3 + 0] * b[0 * 3 + j];
            s = s + a[i * 3 + 1] * b[1 * 3 + j];
            s = s + a[i * 3 + 2] * b[2 * 3 + j];
            c[i * 3 + j] = s;
        }
}

// src/dst: 4 points (x, y) each, already exactly representable in float32.
__device__ static bool homography4(const double src[8], const double dst[8], double H[9])
{
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
    for (int i = 0; i < 4; ++i) { cmx += dst[2 * i]; cmy += dst[2 * i + 1]; cMx += src[2 * i]; cMy += src[2 * i + 1]; }
    cmx /= 4; cmy /= 4; cMx /= 4; cMy /= 4;
    double smx = 0, smy = 0, sMx = 0, sMy = 0;
    for (int i = 0; i < 4; ++i) {
        smx += fabs(dst[2 * i] - cmx); smy += fabs(dst[2 * i + 1] - cmy);
        sMx += fabs(src[2 * i] - cMx); sMy += fabs(src[2 * i + 1] - cMy);
    }
    const double eps = 2.220446049250313e-16;   // DBL_EPSILON
    if (fabs(smx) < eps || fabs(smy) < eps || fabs(sMx) < eps || fabs(sMy) < eps) return false;
    smx = 4 / smx; smy = 4 / smy; sMx = 4 / sMx; sMy = 4 / sMy;
    const double invHnorm[9] = { 1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1 };
    const double Hnorm2[9] = { sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1 };
    // the 4-point problem in normalised coordinates, in closed form: corners arrive as TL, TR, BL, BR; the unit square's cyclic order is
    // points 0, 1, 3, 2
    double nm[4][2], nM[4][2];
    for (int i = 0; i < 4; ++i) {
        nm[i][0] = (dst[2 * i] - cmx) * smx; nm[i][1] = (dst[2 * i + 1] - cmy) * smy;
        nM[i][0] = (src[2 * i] - cMx) * sMx; nM[i][1] = (src[2 * i + 1] - cMy) * sMy;
    }
    double Sd[9], Ss[9], As[9], H0[9];
    const bool okd = square_to_quad(nm[0], nm[1], nm[3], nm[2], Sd), oks = square_to_quad(nM[0], nM[1], nM[3], nM[2], Ss);
    if (!okd || !oks) return false;
    adjugate3(Ss, As);
    // Either quad with ANY three corners collinear (not only the three square_to_quad divides by) makes its square -> quad map singular:
    // det = first row . first column of the adjugate; the normalised coordinates are O(1), so 1e-10 is "no area" (round 1-3's 8 x 8
    // elimination met a zero pivot there).  Same operations in the oracles: the same decision.
    const double det_s = (Ss[0] * As[0] + Ss[1] * As[3]) + Ss[2] * As[6];
    const double det_d = (Sd[0] * (Sd[4] - Sd[5] * Sd[7]) + Sd[1] * (Sd[5] * Sd[6] - Sd[3])) + Sd[2] * (Sd[3] * Sd[7] - Sd[4] * Sd[6]);
    if (!(fabs(det_s) > 1e-10) || !(fabs(det_d) > 1e-10)) return false;
    matmul3(Sd, As, H0);
    double Ht[9], Hd[9];
    matmul3(invHnorm, H0, Ht);
    matmul3(Ht, Hnorm2, Hd);
    if (Hd[8] == 0.0) return false;
    const double sc = 1.0 / Hd[8];
    for (int i = 0; i < 9; ++i) H[i] = Hd[i] * sc;
    return true;
}

// cv::invert 3x3 closed form (what cv2.warpPerspective applies to the forward homography).
__device__ static void invert3x3(const double S[9], double t[9])
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

// Conservative box of the pixels where the cell's warped mask can be non-zero: the forward image of the
// rect dilated by one pixel is a convex quad when the forward denominator is positive on its corners, and
// every pixel passing the mask test maps (projectively) into that dilated rect; +-2 px of slack covers the
// rounding of M = inverse(H_fwd).  Anything odd falls back to the whole frame.
__device__ static void cell_bbox(const double Hf[9], const double M[9], double L, double T, double Rt, double B,
                                 int W, int H, double bbox[4])
{
    const double qx[4] = { L - 1, Rt + 1, L - 1, Rt + 1 };
    const double qy[4] = { T - 1, T - 1, B + 1, B + 1 };
    const double fx[4] = { 0, (double)(W - 1), 0, (double)(W - 1) };
    const double fy[4] = { 0, 0, (double)(H - 1), (double)(H - 1) };
    bool ok = true;
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    for (int i = 0; i < 4 && ok; ++i) {
        const double den = (Hf[6] * qx[i] + Hf[7] * qy[i]) + Hf[8];
        const double wf = (M[6] * fx[i] + M[7] * fy[i]) + M[8];
        if (!(den > 1e-3) || !(wf > 1e-3)) { ok = false; break; }
        const double px = ((Hf[0] * qx[i] + Hf[1] * qy[i]) + Hf[2]) / den;
        const double py = ((Hf[3] * qx[i] + Hf[4] * qy[i]) + Hf[5]) / den;
        if (!(fabs(px) < 1e9) || !(fabs(py) < 1e9)) { ok = false; break; }
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
    if (x0 > x1 || y0 > y1) { bbox[0] = 1; bbox[1] = 1; bbox[2] = 0; bbox[3] = 0; return; }
    bbox[0] = x0; bbox[1] = y0; bbox[2] = x1; bbox[3] = y1;
}

// (16-byte stores to addresses that are only element-aligned: the edge sets start behind an odd number of 8-byte boxes for odd n R C)
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(64) void cell_table_kernel(const double* __restrict__ unstab,
                                                        const double* __restrict__ stab, int n, int W, int H, int R,
                                                        int C, double* __restrict__ records,
                                                        CellBox* __restrict__ boxes, float* __restrict__ edges, float* __restrict__ uedges,
                                                        int32_t* __restrict__ reach, int32_t* __restrict__ grid,
                                                        int32_t* __restrict__ crop, int32_t* __restrict__ status, int32_t* __restrict__ bounds)
{
    const int ncell = R * C;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n) {                         // per-frame crop defaults, meshflowstabilizer.py:992-995
        crop[4 * gid + 0] = 0; crop[4 * gid + 1] = 0; crop[4 * gid + 2] = W - 1; crop[4 * gid + 3] = H - 1;
    }
    if (gid < 4 && bounds) bounds[gid] = gid < 2 ? 0 : (gid == 2 ? W - 1 : H - 1);                      // clip-level defaults (mfs.py:992-995, 1103-1106)
    if (gid <= C) grid[gid] = (int32_t)ceil((double)(W - 1) * ((double)gid / (double)C));             // vertex x
    if (gid <= R) grid[C + 1 + gid] = (int32_t)ceil((double)(H - 1) * ((double)gid / (double)R));     // vertex y
    // One wavefront = 64 consecutive cells of ONE frame (a frame takes reach_parts(R, C) wavefronts), so that the frame's reach is a
    // plain per-wavefront maximum: no atomics, nothing to zero beforehand.  Lanes past the frame's last cell repeat it (identical
    // values to identical addresses) so that the whole wavefront stays active for the cross-lane reduction at the end.
    const int wpf = reach_parts(R, C);
    const int f = (int)(blockIdx.x / (unsigned)wpf), part = (int)(blockIdx.x % (unsigned)wpf);
    if (f >= n) return;                    // (a launch is padded to two wavefronts when n * wpf == 1: the grid needs 65 threads)
    const bool live = part * 64 + (int)threadIdx.x < ncell;
    const int k = live ? part * 64 + (int)threadIdx.x : ncell - 1;
    const long long cid = (long long)f * ncell + k;
    const int r = k / C, c = k % C;
    const size_t vbase = (size_t)f * (R + 1) * (C + 1);
    double ub[8], sb[8];
    for (int q = 0; q < 4; ++q) {          // TL, TR, BL, BR
        const int rr = r + (q >> 1), cc = c + (q & 1);
        const size_t v = vbase + (size_t)rr * (C + 1) + cc;
        const double gx = (double)(float)ceil((double)(W - 1) * ((double)cc / (double)C));
        const double gy = (double)(float)ceil((double)(H - 1) * ((double)rr / (double)R));
        ub[2 * q] = gx; ub[2 * q + 1] = gy;
        // stabilized vertex = grid + (stabilized - unstabilized displacement) in float64, then float32
        sb[2 * q] = (double)(float)(gx + (stab[2 * v] - unstab[2 * v]));
        sb[2 * q + 1] = (double)(float)(gy + (stab[2 * v + 1] - unstab[2 * v + 1]));
    }
    double Hf[9], Hi[9];
    const bool ok = homography4(ub, sb, Hf) && homography4(sb, ub, Hi);
    // The wavefront's 64 records (16 KB), then its edge sets (4 KB + 3 KB), are put together in LDS -- one slot per lane, odd strides:
    // no bank conflicts -- and written out as whole 1 KB runs (a lane storing its own 256-byte record touches 64 different cache lines
    // per store instruction: the table kernel was bound by exactly that, 105 -> 50 us at config 3).
    __shared__ double s_rec[64 * (MF_CELL_DOUBLES + 1)];
    double* rec = &s_rec[threadIdx.x * (MF_CELL_DOUBLES + 1)];
    for (int i = 0; i < MF_CELL_DOUBLES; ++i) rec[i] = 0.0;
    const double L = floor(fmin(fmin(ub[0], ub[2]), fmin(ub[4], ub[6])));
    const double Rt = ceil(fmax(fmax(ub[0], ub[2]), fmax(ub[4], ub[6])));
    const double T = floor(fmin(fmin(ub[1], ub[3]), fmin(ub[5], ub[7])));
    const double B = ceil(fmax(fmax(ub[1], ub[3]), fmax(ub[5], ub[7])));
    rec[MF_CELL_OFF_RECT + 0] = L; rec[MF_CELL_OFF_RECT + 1] = T;
    rec[MF_CELL_OFF_RECT + 2] = Rt; rec[MF_CELL_OFF_RECT + 3] = B;
    CellBox box;
    int rxlo = 0, rylo = 0, rxhi = 0, ryhi = 0;      // how far this cell's box reaches beyond its grid rect
    bool has_reach = false;
    float edv[MF_EDGE_FLOATS], uedv[MF_UEDGE_FLOATS];           // (registers: LDS is busy with the records until they are out)
    float* const ed = edv;
    float* const ued = uedv;
    if (!ok) {
        for (int e = 0; e < 4; ++e) {                                     // never a candidate
            ed[3 * e] = ued[3 * e] = 0.0f; ed[3 * e + 1] = ued[3 * e + 1] = 0.0f; ed[3 * e + 2] = ued[3 * e + 2] = -1e30f; ed[12 + e] = 1.0f;
        }
        rec[MF_CELL_OFF_STATUS] = 1.0;
        rec[MF_CELL_OFF_BBOX + 0] = 1; rec[MF_CELL_OFF_BBOX + 1] = 1; rec[MF_CELL_OFF_BBOX + 2] = 0; rec[MF_CELL_OFF_BBOX + 3] = 0;
        box.x0 = 1; box.y0 = 1; box.x1 = 0; box.y1 = 0;
        if (live) atomicAdd(status, 1);
    } else {
        double M[9], bb[4];
        invert3x3(Hf, M);
        cell_bbox(Hf, M, L, T, Rt, B, W, H, bb);
        for (int i = 0; i < 9; ++i) { rec[MF_CELL_OFF_M + i] = M[i]; rec[MF_CELL_OFF_HI + i] = Hi[i]; }
        for (int i = 0; i < 4; ++i) rec[MF_CELL_OFF_BBOX + i] = bb[i];
        box.x0 = (int16_t)bb[0]; box.y0 = (int16_t)bb[1]; box.x1 = (int16_t)bb[2]; box.y1 = (int16_t)bb[3];
        // Edge functions for the wave-level classification in the warp kernel: with X = 32*Xn/Wd
        // (Xn = M0 x + M1 y + M2, Wd = M6 x + M7 y + M8 > 0 on the frame) the mask test
        // 32(L-1) < rint(X) < 32(Rt+1) is, up to rounding, gL > 0 and gR > 0 with
        //   gL = 32 Xn - (32(L-1) + 1/2) Wd,   gR = (32(Rt+1) - 1/2) Wd - 32 Xn      (same for y),
        // each affine in (x, y), so its extrema over a pixel rectangle sit on the corners.  Stored twice as float32 {a, b, c}:
        // `uedges` as they are (units of 1/32 pixel: what the plan kernel's classification margins are in), and `edges` SCALED by
        // the function's own float32 evaluation error bound m for the warp kernel -- with S = |a| (W-1) + |b| (H-1) + |c|, three
        // rounded coefficients and two fma put fma(a, x, fma(b, y, c)) within 2^-23 S of the exact value anywhere on the frame;
        // m = 1.25 * 2^-23 S and the stored coefficients are a / m, b / m, c / m (the four m follow the twelve coefficients).  So
        // for every edge of every cell the scaled function decides the sign of the exact one whenever its magnitude exceeds 1:
        // one threshold for all, and min() over edges keeps its meaning.
        // A cell whose Wd is not positive on the whole frame gets NaNs: neither "all inside" nor
        // "all outside" can then be concluded and every pixel is tested.
        const double fw = (double)(W - 1), fh = (double)(H - 1);
        const bool regular = (M[6] * 0.0 + M[7] * 0.0) + M[8] > 1e-3 && (M[6] * fw + M[7] * 0.0) + M[8] > 1e-3 &&
                             (M[6] * 0.0 + M[7] * fh) + M[8] > 1e-3 && (M[6] * fw + M[7] * fh) + M[8] > 1e-3;
        const double lox = 32.0 * (L - 1) + 0.5, hix = 32.0 * (Rt + 1) - 0.5;
        const double loy = 32.0 * (T - 1) + 0.5, hiy = 32.0 * (B + 1) - 0.5;
        const double co[12] = {
            32.0 * M[0] - lox * M[6], 32.0 * M[1] - lox * M[7], 32.0 * M[2] - lox * M[8],
            hix * M[6] - 32.0 * M[0], hix * M[7] - 32.0 * M[1], hix * M[8] - 32.0 * M[2],
            32.0 * M[3] - loy * M[6], 32.0 * M[4] - loy * M[7], 32.0 * M[5] - loy * M[8],
            hiy * M[6] - 32.0 * M[3], hiy * M[7] - 32.0 * M[4], hiy * M[8] - 32.0 * M[5] };
        for (int e = 0; e < 4; ++e) {
            const double S = fabs(co[3 * e]) * fw + fabs(co[3 * e + 1]) * fh + fabs(co[3 * e + 2]);
            const double m = 1.25 * 1.1920928955078125e-07 * S, inv = 1.0 / m;
            for (int q = 0; q < 3; ++q) {
                ed[3 * e + q] = regular ? (float)(co[3 * e + q] * inv) : __builtin_nanf("");
                ued[3 * e + q] = regular ? (float)co[3 * e + q] : __builtin_nanf("");
            }
            ed[12 + e] = regular ? (float)m : 1.0f;
        }
        rxlo = (int)L - box.x0; rylo = (int)T - box.y0; rxhi = box.x1 - (int)Rt; ryhi = box.y1 - (int)B;
        has_reach = live && box.x0 <= box.x1;
    }
    boxes[cid] = box;
    {
        const int cells = min(64, ncell - part * 64);          // (lanes past the frame's last cell hold copies: not written)
        const size_t wbase = (size_t)f * ncell + (size_t)part * 64;
        __syncthreads();
        double* __restrict__ grec = records + wbase * MF_CELL_DOUBLES;
#pragma unroll
        for (int pass = 0; pass < MF_CELL_DOUBLES / 2; ++pass) {
            const int flat = pass * 128 + 2 * (int)threadIdx.x, cell = flat / MF_CELL_DOUBLES, i = flat % MF_CELL_DOUBLES;
            if (cell < cells)
                *reinterpret_cast<double2_u*>(grec + flat) = (double2_u){s_rec[cell * (MF_CELL_DOUBLES + 1) + i], s_rec[cell * (MF_CELL_DOUBLES + 1) + i + 1] };
        }
        __syncthreads();
        float* s_ed = reinterpret_cast<float*>(s_rec);
        float* s_ued = s_ed + 64 * (MF_EDGE_FLOATS + 1);
#pragma unroll
        for (int i = 0; i < MF_EDGE_FLOATS; ++i) s_ed[threadIdx.x * (MF_EDGE_FLOATS + 1) + i] = edv[i];
#pragma unroll
        for (int i = 0; i < MF_UEDGE_FLOATS; ++i) s_ued[threadIdx.x * (MF_UEDGE_FLOATS + 1) + i] = uedv[i];
        __syncthreads();
        float* __restrict__ ged = edges + wbase * MF_EDGE_FLOATS;
        float* __restrict__ gued = uedges + wbase * MF_UEDGE_FLOATS;
#pragma unroll
        for (int pass = 0; pass < MF_EDGE_FLOATS / 4; ++pass) {
            const int flat = pass * 256 + 4 * (int)threadIdx.x, cell = flat / MF_EDGE_FLOATS, i = flat % MF_EDGE_FLOATS;
            const float* q = &s_ed[cell * (MF_EDGE_FLOATS + 1) + i];
            if (cell < cells) *reinterpret_cast<float4_u*>(ged + flat) = (float4_u){ q[0], q[1], q[2], q[3] };
        }
#pragma unroll
        for (int pass = 0; pass < MF_UEDGE_FLOATS / 4; ++pass) {
            const int flat = pass * 256 + 4 * (int)threadIdx.x, cell = flat / MF_UEDGE_FLOATS, i = flat % MF_UEDGE_FLOATS;
            const float* q = &s_ued[cell * (MF_UEDGE_FLOATS + 1) + i];
            if (cell < cells) *reinterpret_cast<float4_u*>(gued + flat) = (float4_u){ q[0], q[1], q[2], q[3] };
        }
    }
    // Per-frame maxima of the reach bound the plan kernel's candidate search: this wavefront's share, reduced over its lanes and
    // written to the frame's slot `part` (the plan kernel takes the maximum over the frame's slots).
    int a = has_reach ? rxlo : 0, b = has_reach ? rylo : 0, c2 = has_reach ? rxhi : 0, d = has_reach ? ryhi : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a = max(a, __shfl_xor(a, off)); b = max(b, __shfl_xor(b, off));
        c2 = max(c2, __shfl_xor(c2, off)); d = max(d, __shfl_xor(d, off));
    }
    if (threadIdx.x == 0) {
        int32_t* slot = reach + 4 * ((size_t)f * wpf + part);
        slot[0] = a; slot[1] = b; slot[2] = c2; slot[3] = d;
    }
}

// Footprint plan: one thread per 32 x 8 pixel footprint of the warp kernel.  Lists, in descending cell
// order, the cells that can own a pixel of the footprint, classified with the float32 edge functions at the
// four footprint corners (affine functions: the corners bound the footprint; margin of one unit = 1/32 px):
//   OUT   some edge function is < -1 at all four corners            -> not listed
//   IN    every edge function is > +1 at all four corners           -> listed, ends the list (it owns the rest)
//   MIXED otherwise                                                 -> listed, the warp kernel tests per pixel
// Doing this here costs one lane per footprint instead of a whole wavefront per footprint in the warp kernel.
//
// Also the footprint's SOURCE region (mf_common.h, MF_REGION_STAGED): the inverse homography of every listed cell
// maps the footprint rectangle onto a convex quadrilateral (w > 0 at the four corners), so the bounding box of the
// mapped corners, widened by the bilinear footprint and a pixel of slack, holds every tap of every pixel whichever
// listed cell owns it.  If that box fits the staging window and the frame, the warp kernel fetches it once.
// Cells whose grid interval [g[i], g[i+1]], widened by the frame's reach, meets the pixel interval [a, b]: a contiguous index range
// lo..hi of the n intervals (start from the uniform-grid estimate, then walk the exact vertex coordinates: a step or two).
__device__ __forceinline__ void cell_range_1d(const int* g, int n, int extent, int a, int b, int reach_lo, int reach_hi, int& lo, int& hi)
{
    const float scale = (float)n / (float)(extent - 1);
    lo = min(max((int)((float)(a - reach_hi) * scale) - 1, 0), n - 1);
    hi = min(max((int)((float)(b + reach_lo) * scale) + 1, 0), n - 1);
    while (lo > 0 && g[lo] >= a - reach_hi) --lo;                             // make sure the estimate is not too tight
    while (hi < n - 1 && g[hi + 1] <= b + reach_lo) ++hi;
    while (lo < n - 1 && g[lo + 1] < a - reach_hi) ++lo;                      // then tighten exactly
    while (hi > 0 && g[hi] > b + reach_lo) --hi;
}

// Classification + source region of ONE footprint.  `edge_of(k)` yields the 12 float32 edge coefficients of cell k of this
// frame, `hi_of(k, out9)` its inverse homography as float32: from LDS when the workgroup staged its cell rows, else global.
template <typename EdgeOf, typename HiOf, typename MarginOf, typename BoxOf, typename RectOf>
__device__ __forceinline__ void plan_one_footprint(EdgeOf edge_of, HiOf hi_of, MarginOf margin_of, BoxOf box_of, RectOf rect_of, int c_lo, int c_hi, int r_lo, int r_hi,
                                                   int xa, int xb, int ya, int yb, int W, int H, int C, FootPlan& p, FootRegion& region)
{
    // list entries and their edge codes, four 16-bit fields per register pair (entries 4-7 in the second): a dynamically indexed
    // uint16_t[8] would live in scratch memory
    uint64_t ent[2] = { 0, 0 }, codes = 0;
    int cnt = 0;
    bool overflow = false, closed = false;
    const float cxs[2] = { (float)xa, (float)xb }, cys[2] = { (float)ya, (float)yb };
    const float cxm = 0.5f * (cxs[0] + cxs[1]), cxh = 0.5f * (cxs[1] - cxs[0]), cym = 0.5f * (cys[0] + cys[1]), cyh = 0.5f * (cys[1] - cys[0]);
    float umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f;
    bool sane = true;
    float wlo_all = 1e30f, whi_all = -1e30f;                   // denominator range of every listed cell over the footprint
    // largest sum of the magnitudes of a numerator's / the denominator's terms at the far corner over the listed cells (the premises of
    // the warp kernel's cheap coordinate chain, below), and |h6| of the cell listed last: taken while the matrix is in registers
    float sum_nx = 0.0f, sum_ny = 0.0f, sum_w = 0.0f, h6_last = 0.0f;
    bool coded = true;                                         // every MIXED entry has a one- or two-edge code
    // The candidate range comes from the FRAME's reach (the largest overhang of any cell's box over its grid rect); most of its cells
    // have a box that does not meet this footprint.  A first cheap pass keeps the cells whose own box does (descending order kept;
    // up to eight, 16 bits each -- more: the footprint is handed to the warp kernel's generic path like one with more than eight
    // candidates), so that the classification below runs once per cell that can matter instead of once per cell of the range: the
    // lanes of a wavefront then loop 1-2 times instead of 4.
    uint64_t shortlist[2] = { 0, 0 };
    int boxed = 0;
    for (int r = r_hi; r >= r_lo; --r)
        for (int c = c_hi; c >= c_lo; --c) {
            const CellBox b = box_of(r * C + c);
            if (b.x0 <= xb && b.x1 >= xa && b.y0 <= yb && b.y1 >= ya) {
                const uint64_t entry = (uint64_t)(uint32_t)(r * C + c) << (16 * (boxed & 3));
                if (boxed < 4) shortlist[0] |= entry;
                else if (boxed < 8) shortlist[1] |= entry;
                ++boxed;
            }
        }
    overflow = boxed > 8;
#if MF_PLAN_EXP == 2      // (timing-only build, tools/plan_profile.sh: the candidate shortlist alone)
    p.e[0] = (uint16_t)boxed; p.e[1] = (uint16_t)shortlist[0]; p.e[2] = (uint16_t)shortlist[1]; region.flags_origin = 0; region.src_dwords = 0;
    return;
#endif
#if MF_PLAN_EXP == 4      // (timing-only build: at most ONE candidate classified per footprint -- what the candidate loop costs beyond its first trip)
    if (boxed > 1) boxed = 1;
#endif
    for (int step = 0; step < boxed && !closed && !overflow; ++step) {
        const int k = (int)(((step < 4 ? shortlist[0] : shortlist[1]) >> (16 * (step & 3))) & 0xFFFFu);
        {
            const float* ed = edge_of(k);
            bool all_in = true, any_out = false;
            int uncertain = 0, which = 0, which2 = 0;
            for (int e = 0; e < 4; ++e) {
                // extrema of the affine function a x + b y + c over the footprint rectangle sit on its corners: value at the
                // centre -/+ (|a| half width + |b| half height)  (NaN coefficients fail both tests; the float32 rounding of
                // these few operations is ~1e-3 units against the margin of one unit)
                const float mid = __builtin_fmaf(ed[3 * e], cxm, __builtin_fmaf(ed[3 * e + 1], cym, ed[3 * e + 2]));
                const float half = __builtin_fmaf(fabsf(ed[3 * e]), cxh, fabsf(ed[3 * e + 1]) * cyh);
                const float gmin = mid - half, gmax = mid + half;
                all_in = all_in && gmin > 1.0f;
                any_out = any_out || gmax < -1.0f;
                if (!(gmin > 1.0f)) { ++uncertain; which2 = which; which = e; }
            }
            if (any_out) continue;
            if (cnt == 8) { overflow = true; break; }
            const uint32_t code = MF_PLAN_CODES | (uint32_t)(uncertain == 1 ? which : uncertain == 2 ? (8 | which2 | (which << 4)) : 4);
            if (cnt < 4) codes |= (uint64_t)code << (16 * cnt);
            coded = coded && (all_in || uncertain == 1 || uncertain == 2);
            const uint64_t entry = (uint64_t)((uint32_t)k | MF_PLAN_VALID | (all_in ? MF_PLAN_IN : 0u)) << (16 * (cnt & 3));
            if (cnt < 4) ent[0] |= entry; else ent[1] |= entry;
            ++cnt;
            if (all_in) closed = true;
        }
    }
    // Second loop, over the LISTED cells only (round 6): the corner mapping below is the expensive half of a candidate's work and only listed
    // cells need it.  In one loop the lanes of a wavefront reached it at different trip numbers -- the cells come in DESCENDING order, so a
    // footprint inside one cell first dismisses the neighbours below and to the right whose boxes it meets, one to three trips -- and the block
    // ran in nearly every trip for a few lanes each; now all lanes map the corners of their first listed cell together, then of the second.
    // Same cells in the same order, same operations: the same plans (tools/compare_tables.py).
    for (int i = 0; i < cnt && !overflow; ++i) {
        const int k = (int)(((i < 4 ? ent[0] : ent[1]) >> (16 * (i & 3))) & 0xFFFu);
        {
            // source position of the four footprint corners under this cell's inverse homography (float32 is ample:
            // the window keeps 1/16 pixel of slack, the float32 error at coordinates up to 8192 is below 0.01)
            float h[9];
            hi_of(k, h);
            {
                const float ax = fabsf(h[0]) * cxs[1] + fabsf(h[1]) * cys[1] + fabsf(h[2]), ay = fabsf(h[3]) * cxs[1] + fabsf(h[4]) * cys[1] + fabsf(h[5]),
                            aw = fabsf(h[6]) * cxs[1] + fabsf(h[7]) * cys[1] + fabsf(h[8]);
                if (!(ax <= sum_nx)) sum_nx = ax;                   // (a NaN sticks: the comparisons below then fail)
                if (!(ay <= sum_ny)) sum_ny = ay;
                if (!(aw <= sum_w)) sum_w = aw;
                h6_last = fabsf(h[6]);
            }
            // (fused, the y part shared by the two corners of a footprint row: 18 operations for the twelve values instead of 48)
            const float wy[2] = { __builtin_fmaf(h[7], cys[0], h[8]), __builtin_fmaf(h[7], cys[1], h[8]) };
            const float nxy[2] = { __builtin_fmaf(h[1], cys[0], h[2]), __builtin_fmaf(h[1], cys[1], h[2]) };
            const float nyy[2] = { __builtin_fmaf(h[4], cys[0], h[5]), __builtin_fmaf(h[4], cys[1], h[5]) };
            for (int q = 0; q < 4; ++q) {
                const float cx = cxs[q & 1];
                const float w = __builtin_fmaf(h[6], cx, wy[q >> 1]);
                sane = sane && w > 0.25f && w < 4.0f;                     // (NaN fails)
                const float nx = __builtin_fmaf(h[0], cx, nxy[q >> 1]), ny = __builtin_fmaf(h[3], cx, nyy[q >> 1]);
                wlo_all = fminf(wlo_all, w); whi_all = fmaxf(whi_all, w);
                const float iw = __builtin_amdgcn_rcpf(w);            // (1 ulp: the window keeps a sixteenth of a pixel of slack)
                const float u = nx * iw, v = ny * iw;
                umin = fminf(umin, u); umax = fmaxf(umax, u);
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
        }
    }
    for (int i = 0; i < 4; ++i) { p.e[i] = (uint16_t)(ent[0] >> (16 * i)); p.e[4 + i] = (uint16_t)(ent[1] >> (16 * i)); }
#if MF_PLAN_EXP == 3      // (timing-only build: shortlist + classification + corner mapping, no certificates / window)
    region.flags_origin = (uint32_t)(umin + umax + vmin + vmax + wlo_all + whi_all) + (sane ? 1u : 0u) + (coded ? 2u : 0u); region.src_dwords = (uint32_t)codes;
    return;
#endif
    if (!overflow && cnt <= 4)
        for (int i = 0; i < 4; ++i) p.e[4 + i] = (uint16_t)(codes >> (16 * i));      // short list: room for the per-entry edge codes
    bool unit1 = false;                                            // ONE listed cell whose denominator allows the reciprocal guess
    if (cnt == 1 && sane) {
        // one cell owns the footprint: its denominator range over the footprint (corners: wlo_all / whi_all are this cell's) and h6, for
        // the warp kernel's reciprocal guess
        unit1 = wlo_all > 0.52f && whi_all < 1.9f && h6_last <= 0.9f * 2.5e-4f * (wlo_all * wlo_all);
        if (unit1 && closed)
            p.e[1] = (uint16_t)MF_PLAN_UNIT;                       // (float32 evaluation: 1e-6 of error against margins of 4 % and 10 %)
    }
    if (overflow) {
        // more than 8 candidates: hand the whole range to the warp kernel instead
        p.e[0] = (uint16_t)r_lo; p.e[1] = (uint16_t)r_hi; p.e[2] = (uint16_t)c_lo; p.e[3] = (uint16_t)c_hi;
        p.e[4] = p.e[5] = p.e[6] = 0; p.e[7] = (uint16_t)MF_PLAN_OVERFLOW;
    }
    // Two cells that meet inside the footprint, each with ONE uncertain edge g0, g1 (the pair path of the warp kernel): a pixel
    // without owner would fail both, g0 <= m0 and g1 <= m1 (the kernel's float32 error bands of the two edges).  g0 + g1 is affine,
    // so if it exceeds m0 + m1 + 1 at the four corners it does everywhere and every pixel has an owner -- the footprint can be
    // certified like one whose list ends with an IN cell.
    const uint32_t code0 = (uint32_t)codes & 0xFFFFu, code1 = (uint32_t)(codes >> 16) & 0xFFFFu;
    const bool single_ok[2] = { cnt >= 1 && (code0 & 0xCu) == 0, cnt >= 2 && (code1 & 0xCu) == 0 };       // ONE uncertain edge: codes 0-3
    float single_edge0[3] = { 0.0f, 0.0f, 0.0f };                  // the one uncertain edge of the first entry
    bool covered = closed;
    if (!overflow && cnt == 2 && single_ok[0]) {
        const float* e0 = edge_of((int)(p.e[0] & 0xFFFu)) + 3 * (code0 & 3u);
        for (int q = 0; q < 3; ++q) single_edge0[q] = e0[q];
        if (!closed && single_ok[1]) {
            const float* e1 = edge_of((int)(p.e[1] & 0xFFFu)) + 3 * (code1 & 3u);
            const float a = e0[0] + e1[0], b = e0[1] + e1[1], c = e0[2] + e1[2];
            const float gmin = (fminf(a * cxs[0], a * cxs[1]) + fminf(b * cys[0], b * cys[1])) + c;
            // (the kernel's error bands of the two edges, in the same units: their cells and edge numbers are in the list)
            const float m0 = margin_of(p.e[0] & 0xFFF, code0 & 3), m1 = margin_of(p.e[1] & 0xFFF, code1 & 3);
            covered = gmin > m0 + m1 + 1.0f;
        }
    }
    region.flags_origin = 0;
    region.src_dwords = 0;
    // NOFLAG (for the scan-only pass): every pixel's source coordinates stay between the corner values of the listed cells (below),
    // known to 0.01: with u > 1 + 1/16 - 0.01 and u < W - 2 - 1/16 + 0.01 neither |u| < 1 nor |u - (W-1)| < 1 can hold (same for v;
    // mfs.py:1075-1098); a pixel no cell covers sits at (W+1, H+1) and passes none of the tests either.
    // Taps of a pixel at (u, v): columns ix, ix + 1 with ix = rint(32 u) >> 5 in [floor(u - 1/64), floor(u + 1/64)], same for
    // rows.  u and v over the footprint stay between their corner values (ratios of affine functions, w > 0), and the corner
    // values here are float32 evaluations: error below 0.01 at coordinates up to 8192 (a few operations at 2^-24 relative,
    // the reciprocal to 1 ulp).  So a slack of 1/16 pixel covers both (1/64 + 0.01 < 1/16); a whole pixel on larger frames.
    const bool ranged = sane && cnt > 0 && !overflow && umin > -1e6f && vmin > -1e6f && umax < 1e6f && vmax < 1e6f;
    int ix_lo = 0, ix_hi = 0, iy_lo = 0, iy_hi = 0;
    bool inside = false;                          // every tap inside the frame AND no crop flag possible (see `interior` below)
    if (ranged) {
        const float slack = (W <= 8192 && H <= 8192) ? 0.0625f : 1.0f;
        ix_lo = (int)floorf(umin - slack); ix_hi = (int)floorf(umax + slack) + 1;
        iy_lo = (int)floorf(vmin - slack); iy_hi = (int)floorf(vmax + slack) + 1;
        inside = ix_lo >= 1 && ix_hi <= W - 2 && iy_lo >= 1 && iy_hi <= H - 2;
    }
    const bool noflag = inside;
    if (noflag) region.flags_origin = MF_REGION_NOFLAG;
    if (ranged && (W & 3) == 0 && 3 * W >= MF_STAGE_PITCH && H > MF_STAGE_ROWS) {
        const bool whole = xb - xa == MF_FOOT_W - 1 && yb - ya == MF_FOOT_H - 1;
        // ONE listed cell: a pixel it COVERS passes the cell's mask test, i.e. maps (by M = inverse of the forward homography, which
        // agrees with Hi to ~1e-9) more than 1/64 pixel inside the cell's grid rect widened by one pixel -- so its taps lie in columns
        // L-1 .. Rt+1 and rows T-1 .. B+1 whatever the corner values say (the corners of a MIXED cell's footprint may lie far outside
        // what the cell covers).  Pixels the cell does not cover get the border colour and no tap of theirs matters.
        // (not needed for an IN cell whose corner values already lie inside the frame: the hot footprints)
        int tx_lo = ix_lo, tx_hi = ix_hi, ty_lo = iy_lo, ty_hi = iy_hi;
        if (cnt == 1 && !(closed && inside)) {
            int rl, rt, rr, rb;
            rect_of((int)(p.e[0] & 0xFFFu), rl, rt, rr, rb);
            tx_lo = max(tx_lo, rl - 1); tx_hi = min(tx_hi, rr + 1); ty_lo = max(ty_lo, rt - 1); ty_hi = min(ty_hi, rb + 1);
        }
        // STAGED: the window holds every tap position CLAMPED into the frame -- for a footprint whose taps all lie inside the frame
        // that is every tap; for one on the frame border the warp kernel's per-tap path reads the clamped positions and paints the
        // taps outside in the border colour (cv2.remap BORDER_CONSTANT, mfs.py:1063-1069)
        const int cx_lo = min(max(tx_lo, 0), W - 1), cx_hi = min(max(tx_hi, 0), W - 1);
        const int cy_lo = min(max(ty_lo, 0), H - 1), cy_hi = min(max(ty_hi, 0), H - 1);
        // the window's first byte in a frame row: the dword that holds column cx_lo, but no later than PITCH bytes before the row's
        // end (the copy never reads past a row: the last row of the last frame has nothing behind it); columns first_col .. last_col
        // lie completely inside it
        const uint32_t bs = min((3u * (uint32_t)cx_lo) & ~3u, 3u * (uint32_t)W - (uint32_t)MF_STAGE_PITCH);
        const int last_col = (int)((bs + (uint32_t)MF_STAGE_PITCH - 3u) / 3u);
#ifndef MF_NO_BORDER
        // BORDER window (mf_common.h): one candidate, certified denominator, whole footprint, not a deep one.  The taps of covered
        // pixels reach at most ONE pixel outside the frame (tx / ty ranges above: L >= 0, Rt <= W-1); what they reach there is painted
        // into the window by the kernel, which needs the bytes in front of / behind the window rows to be free.
        const uint32_t code0b = (uint32_t)codes & 0xFFFFu;
        const bool one_coded = cnt == 1 && !overflow && (closed || (code0b & 0x3Fu) != 4u);
        if (one_coded && unit1 && whole && tx_lo <= tx_hi && ty_lo <= ty_hi && !(covered && inside)) {
            const bool pl = tx_lo < 0, pr = tx_hi > W - 1, pt = ty_lo < 0, pb = ty_hi > H - 1;
            const int sy0b = pb ? H - MF_STAGE_ROWS : min(cy_lo, H - MF_STAGE_ROWS);
            const bool fit = cx_hi <= last_col && cy_lo >= sy0b && cy_hi <= sy0b + MF_STAGE_ROWS - 1 &&
                             (!pl || (bs == 0u && cx_hi <= 51)) &&                                         // bytes 156..159 of every row are free
                             (!pr || (bs == 3u * (uint32_t)W - (uint32_t)MF_STAGE_PITCH && 3u * (uint32_t)cx_lo >= bs + 3u)) &&   // bytes 0..2 are
                             (!pt || sy0b == 0) && !(pl && pr) && !(pt && pb);
            if (fit) {
                region.flags_origin = MF_REGION_STAGED | MF_REGION_BORDER | (noflag ? MF_REGION_NOFLAG : 0u) | (pl ? MF_REGION_PAINT_LEFT : 0u) | (pr ? MF_REGION_PAINT_RIGHT : 0u) |
                                      (pt ? MF_REGION_PAINT_TOP : 0u) | (pb ? MF_REGION_PAINT_BOTTOM : 0u) | ((uint32_t)sy0b * MF_STAGE_PITCH + bs);
                region.src_dwords = ((uint32_t)sy0b * (3u * (uint32_t)W) + bs) >> 2;
                p.e[1] = (uint16_t)(p.e[1] | MF_PLAN_BORDER);
                return;
            }
        }
#endif
        const int sy0 = min(cy_lo, H - MF_STAGE_ROWS - 1);
        if (cx_hi <= last_col && cy_hi <= sy0 + MF_STAGE_ROWS - 1)
        {
            // DEEP also asks for a whole footprint (all 256 pixels inside the frame): its lanes are then all active.
            // Interior: every tap inside the frame (ix_lo >= 0, ix_hi <= W - 1) and no crop flag possible -- |u| < 1 needs
            // u < 1 but u >= ix_lo + 1/16 - 1/100; |u - (W-1)| < 1 needs u > W - 2 but u < ix_hi - 1/16 + 1/100 (mfs.py:1075-1098).
            const bool interior = whole && inside;
            const bool deep = covered && interior;
            bool certified = false;                  // one of the warp kernel's three certified shapes: hot, pair, multi
            region.flags_origin = MF_REGION_STAGED | (deep ? MF_REGION_DEEP : 0u) | (noflag ? MF_REGION_NOFLAG : 0u) | ((uint32_t)sy0 * MF_STAGE_PITCH + bs);
            region.src_dwords = ((uint32_t)sy0 * (3u * (uint32_t)W) + bs) >> 2;
            // The premises of the warp kernel's cheap coordinate chain (warp.hip, cheap_quotients) for EVERY listed cell on this footprint:
            // no cancellation in the numerators -- the sum of the magnitudes of a numerator's terms (largest at the far corner: x, y >= 0)
            // at most 8 x the numerator, which is u w >= umin wlo for every listed cell -- denominator terms bounded, denominator above
            // 0.52.  One test for the three paths that use it (hot, pair, multi).
            // (the sums were taken in the candidate loop, largest over the listed cells)
            const bool cheap_all = interior && !overflow && cnt <= 4 && umin > 0.0f && vmin > 0.0f && wlo_all > 0.52f &&
                                   sum_nx <= 7.9f * umin * wlo_all && sum_ny <= 7.9f * vmin * wlo_all && sum_w <= 2.45f;
            if (deep && p.e[1] == (uint16_t)MF_PLAN_UNIT && (p.e[0] & (MF_PLAN_VALID | MF_PLAN_IN)) == (MF_PLAN_VALID | MF_PLAN_IN)) {
                // FAST64: no cancellation in the numerators (sum of the terms' magnitudes at most 8 x the value, everywhere on
                // the footprint: the former grows with x and y, the latter is smallest at a corner), denominator terms bounded --
                // the premises of the warp kernel's error bound for its cheap coordinate chain (warp.hip, cell_coords_fast)
                const bool fast64 = cheap_all;
                p.e[1] = (uint16_t)(MF_PLAN_UNIT | MF_PLAN_HOT | (fast64 ? MF_PLAN_FAST64 : 0u));
                certified = true;
            }
            // the pair shape (`covered`: the second cell is IN, or the two single-edge masks overlap across the footprint)
            if (deep && !overflow && cnt == 2 && single_ok[0] && !(p.e[0] & MF_PLAN_IN) && wlo_all > 0.52f && whi_all < 1.9f) {
                // PAIR_VERT: the deciding edge a x + b y + c runs closer to vertical (the warp kernel then transposes its lanes so that a
                // lane's four pixels run ALONG the edge); PAIR_FAST: both cells satisfy the premises of the kernel's cheap coordinate
                // chain -- as MF_PLAN_FAST64
                const bool vert = fabsf(single_edge0[0]) >= fabsf(single_edge0[1]);
                const bool fast = cheap_all;
                p.e[2] = (uint16_t)(MF_PLAN_HOT | (fast ? MF_PLAN_PAIR_FAST : 0u) | (vert ? MF_PLAN_PAIR_VERT : 0u));
                certified = true;
            }
            // the multi shape: coverage is left to the kernel
            else if (interior && !overflow && cnt >= 2 && cnt <= 4 && coded && wlo_all > 0.52f && whi_all < 1.9f) {
                const bool fast = cheap_all;
                p.e[4] = (uint16_t)(p.e[4] | MF_PLAN_HOT | (fast ? MF_PLAN_MULTI_FAST : 0u) | ((uint32_t)(cnt - 1) << MF_PLAN_COUNT_SHIFT));
                certified = true;
            }
#ifndef MF_NO_COMPACT            // (A/B switch shared with warp.hip: both sides must agree on the window layout)
            // COMPACT window for the three certified shapes (hot, pair, multi -- all whole and interior): 9 rows x 112 bytes hold every tap of
            // every pixel whichever listed cell owns it (ix / iy ranges above: the corner values of EVERY listed cell) -> one global->LDS
            // load instead of two, and the conflict-free lane -> row mapping of warp.hip.  (The mesh warp is continuous across cell edges --
            // neighbouring cells share their vertices -- so a footprint on an edge needs no larger window than one inside a cell.)
            const uint32_t cbs = (3u * (uint32_t)ix_lo) & ~3u;
            if (certified && iy_hi - iy_lo + 1 <= MF_COMPACT_ROWS && 3u * (uint32_t)ix_hi + 3u <= cbs + MF_COMPACT_PITCH &&
                iy_lo + MF_COMPACT_ROWS <= H - 1 && cbs + 16u <= 3u * (uint32_t)W) {      // (the load's 64th chunk: the first 16 bytes of
                                                                                        // row sy0 + 9 from byte cbs -- inside that row, hence
                                                                                        // inside the clip even on its last frame)
                region.flags_origin = MF_REGION_STAGED | (deep ? MF_REGION_DEEP : 0u) | MF_REGION_NOFLAG | MF_REGION_COMPACT | ((uint32_t)iy_lo * MF_COMPACT_PITCH + cbs);
                region.src_dwords = ((uint32_t)iy_lo * (3u * (uint32_t)W) + cbs) >> 2;
            }
#endif
        }
    }
}

#ifndef MF_PLAN_PER_THREAD
#define MF_PLAN_PER_THREAD 4
#endif
constexpr int kPlanTile = 256 * MF_PLAN_PER_THREAD;      // footprints per workgroup: each thread plans MF_PLAN_PER_THREAD of them, one staging for all
#ifndef MF_PLAN_STAGE_CELLS
#define MF_PLAN_STAGE_CELLS 256
#endif
constexpr int kPlanStageCells = MF_PLAN_STAGE_CELLS;          // cells (whole mesh rows) a workgroup keeps in LDS: 21 KB

// grid = n * ceil(footprints per frame / kPlanTile): a workgroup of 256 threads handles kPlanTile = 1024 consecutive footprints of ONE
// frame, four per thread one after the other -- a dozen rows of footprints, which only meet a few mesh rows.  Those rows' edge functions
// and inverse homographies are staged in LDS once (the per-footprint loops then run on LDS latency instead of dependent L2 round trips);
// when they do not fit, from global.  (The staging -- three dependent global round trips and three barriers per workgroup -- was a
// quarter of a workgroup's life at one footprint per thread: 1 / 2 / 4 / 6 / 8 per thread: cell table + plan 104.8 / 94.5 / 91.3 / 103 /
// 104 us at config 2, 266 / 248 / 231 / 250 / 397 at config 3; twice the staged cells: slower, 115 / 296.)
// Six wavefronts per SIMD: what the workgroup's 26 KB of LDS admit (six workgroups per CU) -- the kernel needs 85 registers by itself, the cap to
// 80 costs no scratch and, with the candidate loop split in two (plan_one_footprint), is worth 3-5 % of table + plan (round 6:
// 88.5 -> 86.3 (split) -> 83.7 us at config 2, 224.9 -> 219.9 -> 208.3 at config 3; tables byte-identical, tools/compare_tables.py).
#ifndef MF_PLAN_WAVES
#define MF_PLAN_WAVES 6
#endif
#define MF_PLAN_ATTR __attribute__((amdgpu_waves_per_eu(MF_PLAN_WAVES, MF_PLAN_WAVES)))
__global__ __launch_bounds__(256) MF_PLAN_ATTR void footprint_plan_kernel(const float* __restrict__ uedges, const float* __restrict__ edges,
                                                             const double* __restrict__ records, const CellBox* __restrict__ boxes,
                                                             const int32_t* __restrict__ reach,
                                                             const int32_t* __restrict__ grid, int n, int W, int H, int R,
                                                             int C, FootPlan* __restrict__ plan, FootRegion* __restrict__ regions)
{
    __shared__ int s_gx[66], s_gy[66];
    __shared__ float s_edge[kPlanStageCells * MF_UEDGE_FLOATS];
    __shared__ float s_hi[kPlanStageCells * 9];
    __shared__ CellBox s_box[kPlanStageCells];
    if ((int)threadIdx.x <= C) s_gx[threadIdx.x] = grid[threadIdx.x];
    if ((int)threadIdx.x <= R) s_gy[threadIdx.x] = grid[C + 1 + threadIdx.x];
    __syncthreads();
    const int nfx = (W + MF_FOOT_W - 1) / MF_FOOT_W, nfy = (H + MF_FOOT_H - 1) / MF_FOOT_H, per_frame = nfx * nfy;
    const int blocks_per_frame = (per_frame + kPlanTile - 1) / kPlanTile;
    const int f = (int)(blockIdx.x / (unsigned)blocks_per_frame);
    if (f >= n) return;
    const int rem0 = ((int)blockIdx.x - f * blocks_per_frame) * kPlanTile;
    // the frame's reach = maximum over the slots its cell-table wavefronts wrote (at most 64 of them)
    __shared__ int s_reach[4];
    {
        const int wpf = reach_parts(R, C);
        int v = 0;
        if ((int)threadIdx.x < 4 * wpf) v = reach[4 * (size_t)f * wpf + threadIdx.x];      // thread t: slot t / 4, component t % 4
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) v = max(v, __shfl_xor(v, off));              // lanes with equal t % 4 within a wavefront
        if (threadIdx.x < 4) s_reach[threadIdx.x] = 0;
        __syncthreads();
        if ((threadIdx.x & 63) < 4 && (int)(threadIdx.x & ~63u) < 4 * wpf) atomicMax(&s_reach[threadIdx.x & 3], v);
        __syncthreads();
    }
    const int rxlo = s_reach[0], rylo = s_reach[1], rxhi = s_reach[2], ryhi = s_reach[3];
    const float* __restrict__ fedge = uedges + (size_t)f * R * C * MF_UEDGE_FLOATS;
    const float* __restrict__ fmargin = edges + (size_t)f * R * C * MF_EDGE_FLOATS + 12;      // the scaled set's error bounds
    const double* __restrict__ frec = records + (size_t)f * R * C * MF_CELL_DOUBLES;
    const CellBox* __restrict__ fbox = boxes + (size_t)f * R * C;

    // mesh rows the workgroup's footprints can meet (same widening by the frame's reach as per footprint)
    const int ya0 = (rem0 / nfx) * MF_FOOT_H, yb1 = min((min(rem0 + kPlanTile - 1, per_frame - 1) / nfx) * MF_FOOT_H + MF_FOOT_H - 1, H - 1);
    int rb_lo = 0, rb_hi = R - 1;
    while (rb_lo < R - 1 && s_gy[rb_lo + 1] < ya0 - ryhi) ++rb_lo;
    while (rb_hi > 0 && s_gy[rb_hi] > yb1 + rylo) --rb_hi;
    const int staged_cells = (rb_hi - rb_lo + 1) * C;
    const bool staged = rb_hi >= rb_lo && staged_cells <= kPlanStageCells;          // (workgroup-uniform)
    if (staged) {
        const int k0 = rb_lo * C;
        for (int i = threadIdx.x; i < staged_cells * MF_UEDGE_FLOATS; i += 256) s_edge[i] = fedge[(size_t)k0 * MF_UEDGE_FLOATS + i];
        for (int i = threadIdx.x; i < staged_cells; i += 256) s_box[i] = fbox[k0 + i];
        for (int i = threadIdx.x; i < staged_cells * 9; i += 256) {
            const int cell = i / 9, j = i - 9 * cell;
            s_hi[i] = (float)frec[(size_t)(k0 + cell) * MF_CELL_DOUBLES + MF_CELL_OFF_HI + j];
        }
    }
    // candidate cell ranges per footprint column and per footprint row of this workgroup, computed once (a footprint's ranges
    // depend only on its column and on its row) instead of eight data-dependent loops per thread: table + plan -4 % at cfg2
    __shared__ uint32_t s_crange[256], s_rrange[258];
    const int fy_first = rem0 / nfx;
    const int rows_here = min(rem0 + kPlanTile - 1, per_frame - 1) / nfx - fy_first + 1;
    const bool col_table = nfx <= 256;
    if (col_table)
        for (int i = threadIdx.x; i < nfx; i += 256) {
            int lo, hi;
            cell_range_1d(s_gx, C, W, i * MF_FOOT_W, min(i * MF_FOOT_W + MF_FOOT_W - 1, W - 1), rxlo, rxhi, lo, hi);
            s_crange[i] = (uint32_t)lo | (uint32_t)hi << 16;
        }
    const bool row_table = rows_here <= 258;                   // (a tile of a very narrow frame has more rows: ranges per thread then)
    if (row_table)
    for (int i = threadIdx.x; i < rows_here; i += 256) {
        int lo, hi;
        const int yy0 = (fy_first + i) * MF_FOOT_H;
        cell_range_1d(s_gy, R, H, yy0, min(yy0 + MF_FOOT_H - 1, H - 1), rylo, ryhi, lo, hi);
        s_rrange[i] = (uint32_t)lo | (uint32_t)hi << 16;
    }
    __syncthreads();
#if MF_PLAN_EXP == 1      // (timing-only build: launch + staging + range tables alone)
    if (s_crange[threadIdx.x] == 0xFFFFFFFFu && s_edge[threadIdx.x] == 1.5f) plan[blockIdx.x].e[0] = 1;
    return;
#endif
#pragma unroll 1
    for (int part = 0; part < MF_PLAN_PER_THREAD; ++part) {
    const int rem = rem0 + part * 256 + (int)threadIdx.x;
    if (rem >= per_frame) return;
    const int fy = rem / nfx, fx = rem - fy * nfx;
    const int xa = fx * MF_FOOT_W, xb = min(xa + MF_FOOT_W - 1, W - 1);
    const int ya = fy * MF_FOOT_H, yb = min(ya + MF_FOOT_H - 1, H - 1);
    int c_lo, c_hi;
    if (col_table) { c_lo = (int)(s_crange[fx] & 0xFFFFu); c_hi = (int)(s_crange[fx] >> 16); }
    else cell_range_1d(s_gx, C, W, xa, xb, rxlo, rxhi, c_lo, c_hi);
    int r_lo, r_hi;
    if (row_table) { r_lo = (int)(s_rrange[fy - fy_first] & 0xFFFFu); r_hi = (int)(s_rrange[fy - fy_first] >> 16); }
    else cell_range_1d(s_gy, R, H, ya, yb, rylo, ryhi, r_lo, r_hi);
    FootPlan p;
    FootRegion region;
    if (staged) {
        const int k0 = rb_lo * C;
        plan_one_footprint([&](int k) { return (const float*)&s_edge[(k - k0) * MF_UEDGE_FLOATS]; },
                           [&](int k, float (&h)[9]) { for (int j = 0; j < 9; ++j) h[j] = s_hi[(k - k0) * 9 + j]; },
                           [&](int k, int e) { return fmargin[(size_t)k * MF_EDGE_FLOATS + e]; },
                           [&](int k) { return s_box[k - k0]; },
                           [&](int k, int& rl, int& rt, int& rr, int& rb) { const int r = k / C, c = k - r * C; rl = s_gx[c]; rr = s_gx[c + 1]; rt = s_gy[r]; rb = s_gy[r + 1]; },
                           c_lo, c_hi, r_lo, r_hi, xa, xb, ya, yb, W, H, C, p, region);
    } else {
        plan_one_footprint([&](int k) { return fedge + (size_t)k * MF_UEDGE_FLOATS; },
                           [&](int k, float (&h)[9]) {
                               const double* __restrict__ hi = frec + (size_t)k * MF_CELL_DOUBLES + MF_CELL_OFF_HI;
                               for (int j = 0; j < 9; ++j) h[j] = (float)hi[j];
                           },
                           [&](int k, int e) { return fmargin[(size_t)k * MF_EDGE_FLOATS + e]; },
                           [&](int k) { return fbox[k]; },
                           [&](int k, int& rl, int& rt, int& rr, int& rb) { const int r = k / C, c = k - r * C; rl = s_gx[c]; rr = s_gx[c + 1]; rt = s_gy[r]; rb = s_gy[r + 1]; },
                           c_lo, c_hi, r_lo, r_hi, xa, xb, ya, yb, W, H, C, p, region);
    }
    const size_t gid = (size_t)f * per_frame + rem;
    plan[gid] = p;
    regions[gid] = region;
    }
}

int launch_cell_table(const double* unstab, const double* stab, int n, int W, int H, int R, int C,
                      const TableView& tv, int32_t* crop, int32_t* status, hipStream_t st, bool first_of_table)
{
    if (n <= 0 || R <= 0 || C <= 0 || W < 2 || H < 2 || W > 32767 || H > 32767 || R > 64 || C > 64) {
        set_error("mf_cell_table_f64: unsupported shape n=%d W=%d H=%d R=%d C=%d", n, W, H, R, C);
        return MF_ERR_INVALID_ARG;
    }
    // reach_parts(R, C) wavefronts per frame (64 cells each); the grid also initialises the n crop rows and the R + C + 2 vertex
    // coordinates by global thread index: at least max(n, 65) threads (R, C <= 64)
    long long blocks = (long long)n * reach_parts(R, C);
    if (blocks * 64 < n) blocks = (n + 63) / 64;              // (cannot happen: every frame has a wavefront; kept as a guard)
    if (blocks < 2) blocks = 2;
    hipLaunchKernelGGL(cell_table_kernel, dim3((unsigned)blocks), dim3(64), 0, st, unstab, stab, n, W, H, R, C, tv.records, tv.boxes,
                       tv.edges, tv.uedges, tv.reach, tv.grid, crop, status, first_of_table ? tv.bounds : nullptr);
    int rc = hip_fail(hipGetLastError(), "cell_table_kernel launch");
    if (rc != MF_OK) return rc;
    const size_t per_frame = plan_count(1, W, H);
    hipLaunchKernelGGL(footprint_plan_kernel, dim3((unsigned)(((per_frame + kPlanTile - 1) / kPlanTile) * (size_t)n)), dim3(256), 0, st, tv.uedges, tv.edges, tv.records, tv.boxes,
                       tv.reach, tv.grid, n, W, H, R, C, tv.plan, tv.regions);
    return hip_fail(hipGetLastError(), "footprint_plan_kernel launch");
}

}  // namespace mf

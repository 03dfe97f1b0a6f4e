// Kernel 2b: per-frame, per-cell mesh warp with the crop-boundary scan folded in.
//
// Reference (meshflowstabilizer.py, "mfs.py"): for every frame, every mesh cell in row-major order warps a
// full-frame float64 mask (cv2.warpPerspective, mfs.py:1050-1052), transforms ALL pixel coordinates with
// the cell's inverse homography (cv2.perspectiveTransform, mfs.py:1054) and merges them into the frame's
// coordinate map under that mask (np.where, mfs.py:1060-1061); one cv2.remap (bilinear, constant border,
// mfs.py:1063-1069) and four edge scans (mfs.py:1075-1098) follow.  O(R*C*H*W) per frame.
//
// This kernel computes the same result in one pass over output pixels:
//   owner(pixel) = LAST cell in row-major order whose mask test passes  (painter merge)
//   (u, v)       = float32( Hi_owner * (x, y, 1) ), or (W+1, H+1) when no cell covers the pixel
//   crop flags   = |u| < 1, |u-(W-1)| < 1, |v| < 1, |v-(H-1)| < 1  -> atomic max/min per frame
//   out          = cv2.remap fixed-point bilinear of the source frame at (u, v)
// Coordinates are float64 exactly as OpenCV evaluates them (no FMA contraction: -ffp-contract=off), then
// float32, then 1/32-pixel fixed point; interpolation is integer.  The result is bit-identical to the CPU
// oracle (oracle/warp_oracle.c).
//
// Mapping (gfx950): a 256-thread workgroup owns a 128 x 16 pixel tile of one frame (tile rows start on a
// 128-byte boundary of the 3-byte-per-pixel output when W % 128 == 0, as for 1080p and 4K).  Each
// wavefront owns a 32-pixel-wide column of the tile and walks it in two 32 x 8 footprints; a lane owns 4
// consecutive pixels of one row (12 contiguous output bytes -> one global_store_dwordx3).  Per wavefront:
//   1. candidate cells = those whose conservative box (compact int16 array, coalesced 8-byte loads)
//      touches the wave's 32 x 16 region, collected with __ballot into wave-uniform 64-bit masks;
//   2. candidates are visited in DESCENDING cell order; the cell index is wave-uniform, so its record
//      (M, rect, Hi) is fetched with scalar loads into SGPRs -- nothing per-cell is held per lane or in
//      LDS; the first cell whose mask test passes owns the pixel; the loop ends when every pixel is owned;
//   3. source taps are fetched with unaligned 4-byte loads (3 bytes used), served by L1/L2: neighbouring
//      lanes touch neighbouring bytes because the motion is a few pixels.
// Algorithmic HBM traffic: 2*H*W*3 bytes per frame (each source byte read once, each output byte written
// once); the cell table adds R*C*264 bytes per frame (< 1.5 %).  No dense contraction: no MFMA.
#include "mf_common.h"

namespace mf {

constexpr int TILE_W = 128;
constexpr int TILE_H = 16;
constexpr int WAVE_W = 32;      // pixels per wavefront footprint row (8 lanes x 4 pixels)
constexpr int FOOT_H = 8;       // rows per footprint (64 lanes / 8)
constexpr int QUADS = TILE_H / FOOT_H;   // footprints per wavefront
constexpr int NPIX = 4 * QUADS;          // pixels per lane

__device__ __forceinline__ int cv_round_f32(float v)
{
    const float r = rintf(v);
    return (r >= -2147483648.0f && r < 2147483648.0f) ? (int)r : (int)0x80000000;
}

// One source pixel as B | G << 8 | R << 16, or the border colour when (tx, ty) is outside the frame.
// `limit` = bytes from the frame base to the end of the whole frame stack, so the 4-byte load of the
// very last pixel is shifted back by one byte instead of running past the allocation.
__device__ __forceinline__ uint32_t fetch_bgr(const uint8_t* __restrict__ frame, int W, int H, int tx, int ty,
                                              uint32_t border, size_t limit)
{
    if ((unsigned)tx < (unsigned)W && (unsigned)ty < (unsigned)H) {
        const size_t o = ((size_t)ty * W + tx) * 3;
        uint32_t v;
        if (o + 4 <= limit) {
            __builtin_memcpy(&v, frame + o, 4);
        } else {
            __builtin_memcpy(&v, frame + o - 1, 4);
            v >>= 8;
        }
        return v & 0xFFFFFFu;
    }
    return border;
}

__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ frames, uint8_t* __restrict__ out,
                                                   const double* __restrict__ records,
                                                   const CellBox* __restrict__ boxes, int n, int W, int H,
                                                   int ncell, uint32_t border, int32_t* __restrict__ crop)
{
    const int f = blockIdx.z;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * TILE_W + wave * WAVE_W + (lane & 7) * 4;   // first of this lane's 4 pixels
    const int ybase = blockIdx.y * TILE_H + (lane >> 3);                  // row of quad 0; quad q adds 8*q
    const int rx0 = blockIdx.x * TILE_W + wave * WAVE_W;                  // wave region (inclusive)
    if (rx0 >= W) return;                                                 // whole wave outside the frame
    const int rx1 = min(rx0 + WAVE_W - 1, W - 1);
    const int ry0 = blockIdx.y * TILE_H;
    const int ry1 = min(ry0 + TILE_H - 1, H - 1);

    const size_t frame_bytes = (size_t)W * H * 3;
    const uint8_t* __restrict__ src = frames + (size_t)f * frame_bytes;
    uint8_t* __restrict__ dst = out + (size_t)f * frame_bytes;
    const size_t limit = (size_t)(n - f) * frame_bytes;
    const double* __restrict__ frec = records + (size_t)f * ncell * MF_CELL_DOUBLES;
    const CellBox* __restrict__ fbox = boxes + (size_t)f * ncell;

    // Per-pixel state.  A pixel outside the frame counts as owned from the start.
    int sx[NPIX], sy[NPIX];
    uint32_t unowned = 0;
#pragma unroll
    for (int q = 0; q < QUADS; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = q * 4 + j;
            sx[p] = (W + 1) * 32;          // cvRound(float(W+1)*32): the "uncovered" default, mfs.py:983-984
            sy[p] = (H + 1) * 32;
            if (x0 + j < W && ybase + q * FOOT_H < H) unowned |= 1u << p;
        }
    int c_left = 0, c_top = 0, c_right = W - 1, c_bottom = H - 1;

    const int nchunk = (ncell + 63) >> 6;
    bool done = __ballot(unowned != 0) == 0;
    for (int ch = nchunk - 1; ch >= 0 && !done; --ch) {
        const int kc = ch * 64 + lane;
        bool hit = false;
        if (kc < ncell) {
            const CellBox b = fbox[kc];
            hit = b.x0 <= b.x1 && b.x1 >= rx0 && b.x0 <= rx1 && b.y1 >= ry0 && b.y0 <= ry1;
        }
        unsigned long long cand = __ballot(hit);
        while (cand != 0 && !done) {
            const int bit = 63 - __clzll(cand);
            cand &= ~(1ull << bit);
            const int k = ch * 64 + bit;                                   // wave-uniform
            const double* __restrict__ rec = frec + (size_t)k * MF_CELL_DOUBLES;
            double M[9], Hi[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { M[i] = rec[MF_CELL_OFF_M + i]; Hi[i] = rec[MF_CELL_OFF_HI + i]; }
            const int lo_x = 32 * ((int)rec[MF_CELL_OFF_RECT + 0] - 1);
            const int lo_y = 32 * ((int)rec[MF_CELL_OFF_RECT + 1] - 1);
            const int hi_x = 32 * ((int)rec[MF_CELL_OFF_RECT + 2] + 1);
            const int hi_y = 32 * ((int)rec[MF_CELL_OFF_RECT + 3] + 1);
#pragma unroll
            for (int q = 0; q < QUADS; ++q) {
                if (((unowned >> (4 * q)) & 15u) == 0) continue;
                const int y = ybase + q * FOOT_H;
                const double yy = (double)y;
                // cv2.warpPerspective evaluates destination pixels in 64-wide blocks (imgwarp.cpp):
                //   X0 = M0*xb + M1*y + M2 at the block start xb, then X0 + M0*x1 inside the block.
                const double xb = (double)(x0 & ~63);
                const double X0 = (M[0] * xb + M[1] * yy) + M[2];
                const double Y0 = (M[3] * xb + M[4] * yy) + M[5];
                const double W0 = (M[6] * xb + M[7] * yy) + M[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = q * 4 + j;
                    if (!((unowned >> p) & 1u)) continue;
                    const int x = x0 + j;
                    const double x1 = (double)(x & 63);
                    const double Wd = W0 + M[6] * x1;
                    const double Ws = Wd != 0.0 ? 32.0 / Wd : 0.0;
                    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * Ws));
                    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * Ws));
                    const int X = (int)rint(fX);
                    const int Y = (int)rint(fY);
                    // non-zero bilinear sample of the 255-filled rect <=> a tap with non-zero weight on it
                    if (!(X > lo_x && X < hi_x && Y > lo_y && Y < hi_y)) continue;
                    unowned &= ~(1u << p);
                    // cv2.perspectiveTransform (matmul.simd.hpp): float32 point, float64 matrix
                    const double xs = (double)x;
                    double w = (xs * Hi[6] + yy * Hi[7]) + Hi[8];
                    float u = 0.0f, v = 0.0f;
                    if (fabs(w) > 1.1920928955078125e-07) {
                        w = 1.0 / w;
                        u = (float)(((xs * Hi[0] + yy * Hi[1]) + Hi[2]) * w);
                        v = (float)(((xs * Hi[3] + yy * Hi[4]) + Hi[5]) * w);
                    }
                    // crop-boundary scan, mfs.py:1075-1098 (|u - e| < 1 on exact values)
                    if (u > -1.0f && u < 1.0f) c_left = max(c_left, x);
                    if (u > (float)(W - 2) && u < (float)W) c_right = min(c_right, x);
                    if (v > -1.0f && v < 1.0f) c_top = max(c_top, y);
                    if (v > (float)(H - 2) && v < (float)H) c_bottom = min(c_bottom, y);
                    // cv2.remap: 1/32-pixel fixed point, round half to even
                    sx[p] = cv_round_f32(u * 32.0f);
                    sy[p] = cv_round_f32(v * 32.0f);
                }
            }
            done = __ballot(unowned != 0) == 0;
        }
    }

    // Crop bounds: wave reduction, then at most one atomic per bound per wave (most waves have none).
    {
        const bool any = c_left != 0 || c_top != 0 || c_right != W - 1 || c_bottom != H - 1;
        if (__ballot(any) != 0) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                c_left = max(c_left, __shfl_xor(c_left, off));
                c_top = max(c_top, __shfl_xor(c_top, off));
                c_right = min(c_right, __shfl_xor(c_right, off));
                c_bottom = min(c_bottom, __shfl_xor(c_bottom, off));
            }
            if (lane == 0) {
                if (c_left != 0) atomicMax(&crop[4 * f + 0], c_left);
                if (c_top != 0) atomicMax(&crop[4 * f + 1], c_top);
                if (c_right != W - 1) atomicMin(&crop[4 * f + 2], c_right);
                if (c_bottom != H - 1) atomicMin(&crop[4 * f + 3], c_bottom);
            }
        }
    }

    // Bilinear gather + store.
    const bool fast_store = (W & 3) == 0;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
        const int y = ybase + q * FOOT_H;
        if (y >= H || x0 >= W) continue;
        uint32_t px[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = q * 4 + j;
            const int ix = max(-32768, min(32767, sx[p] >> 5));
            const int iy = max(-32768, min(32767, sy[p] >> 5));
            const int fx = sx[p] & 31, fy = sy[p] & 31;
            if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) { px[j] = border; continue; }
            const uint32_t p00 = fetch_bgr(src, W, H, ix, iy, border, limit);
            const uint32_t p01 = fetch_bgr(src, W, H, ix + 1, iy, border, limit);
            const uint32_t p10 = fetch_bgr(src, W, H, ix, iy + 1, border, limit);
            const uint32_t p11 = fetch_bgr(src, W, H, ix + 1, iy + 1, border, limit);
            const uint32_t w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
            uint32_t r = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t a = w00 * ((p00 >> (8 * c)) & 255u) + w01 * ((p01 >> (8 * c)) & 255u) +
                                   w10 * ((p10 >> (8 * c)) & 255u) + w11 * ((p11 >> (8 * c)) & 255u);
                r |= ((a + 512u) >> 10) << (8 * c);      // == (a*32 + 2^14) >> 15
            }
            px[j] = r;
        }
        const size_t o = ((size_t)y * W + x0) * 3;
        if (fast_store && x0 + 3 < W) {
            uint3 d;
            d.x = px[0] | (px[1] << 24);
            d.y = (px[1] >> 8) | (px[2] << 16);
            d.z = (px[2] >> 16) | (px[3] << 8);
            *reinterpret_cast<uint3*>(dst + o) = d;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < W) {
                    dst[o + 3 * j + 0] = (uint8_t)(px[j]);
                    dst[o + 3 * j + 1] = (uint8_t)(px[j] >> 8);
                    dst[o + 3 * j + 2] = (uint8_t)(px[j] >> 16);
                }
        }
    }
}

int launch_warp(const uint8_t* frames, uint8_t* out, const double* records, const CellBox* boxes, int n, int W,
                int H, int R, int C, uint32_t border, int32_t* crop, hipStream_t st)
{
    if (n <= 0 || n > 65535 || W < 2 || H < 2 || W > 32767 || H > 32767 || R <= 0 || C <= 0 || R * C > 4096) {
        set_error("mf_warp_u8c3: unsupported shape n=%d W=%d H=%d R=%d C=%d", n, W, H, R, C);
        return MF_ERR_INVALID_ARG;
    }
    const dim3 grid((W + TILE_W - 1) / TILE_W, (H + TILE_H - 1) / TILE_H, n);
    hipLaunchKernelGGL(warp_kernel, grid, dim3(256), 0, st, frames, out, records, boxes, n, W, H, R * C, border, crop);
    return hip_fail(hipGetLastError(), "warp_kernel launch");
}

// Clip-level bounds, mfs.py:1103-1106.
__global__ __launch_bounds__(256) void crop_reduce_kernel(const int32_t* __restrict__ crop, int n, int W, int H,
                                                          int32_t* __restrict__ bounds)
{
    __shared__ int32_t red[4][256];
    int l = 0, t = 0, r = W - 1, b = H - 1;
    for (int i = threadIdx.x; i < n; i += 256) {
        l = max(l, crop[4 * i + 0]); t = max(t, crop[4 * i + 1]);
        r = min(r, crop[4 * i + 2]); b = min(b, crop[4 * i + 3]);
    }
    red[0][threadIdx.x] = l; red[1][threadIdx.x] = t; red[2][threadIdx.x] = r; red[3][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = max(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = min(red[3][threadIdx.x], red[3][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) bounds[threadIdx.x] = red[threadIdx.x][0];
}

int launch_crop_reduce(const int32_t* crop, int n, int W, int H, int32_t* bounds, hipStream_t st)
{
    if (n <= 0) { set_error("mf_crop_reduce: n=%d", n); return MF_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(crop_reduce_kernel, dim3(1), dim3(256), 0, st, crop, n, W, H, bounds);
    return hip_fail(hipGetLastError(), "crop_reduce_kernel launch");
}

}  // namespace mf

// Crop + bilinear resize: the step right after the warp in the reference's stabilize()
// (meshflowstabilizer.py:159 -> _crop_frames, :1111-1157): every frame is cropped to the clip-level bounds
// (inclusive) and scaled back to (W, H) with cv2.resize's default INTER_LINEAR.
//
// cv2.resize for 8-bit images (imgproc/resize.cpp, cv::hal::resize + resizeGeneric_) is a two-pass 11-bit
// fixed-point interpolation:
//   scale = 1 / ((double)dst / src);  f = float((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s
//   x axis: s < 0 -> (0, 0);  s >= src-1 -> (src-1, 0)        y axis: the two row indices are clipped instead
//   weights  a0 = cvRound((1 - f) * 2048), a1 = cvRound(f * 2048)          (int16)
//   horizontal  t  = S[s] * a0 + S[s+1] * a1                                (int32)
//   vertical    out = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2
// resize_tables_kernel builds the per-column / per-row tables (W + H entries) on the device in the same
// float / double operations; resize_kernel applies them: a lane owns 4 consecutive output pixels (one
// 12-byte store), the two source rows of an output row staged in LDS, v_dot2_u32_u16 for the horizontal pass.
// Memory-bound in principle (2*H*W*3 bytes per frame).
#include "mf_common.h"

namespace mf {

// x: ofs = sx, w = 16 a0 | 16 a1 << 16 (pre-scaled, see resize_kernel);  y: ofs = sy0 | sy1 << 16, w = b0 | b1 << 16
struct ResizeTab { int32_t ofs; uint32_t w; };

__device__ __forceinline__ int cv_round_pos(float v) { return (int)rintf(v); }

__global__ __launch_bounds__(256) void resize_tables_kernel(int cw, int ch, int W, int H, double scale_x, double scale_y,
                                                            ResizeTab* __restrict__ xtab, ResizeTab* __restrict__ ytab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W) {
        float fx = (float)(((double)i + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.0f; sx = 0; }
        if (sx >= cw - 1) { fx = 0.0f; sx = cw - 1; }
        const int a0 = cv_round_pos((1.0f - fx) * 2048.0f), a1 = cv_round_pos(fx * 2048.0f);
        xtab[i].ofs = sx;
        xtab[i].w = ((uint32_t)a0 << 4) | ((uint32_t)a1 << 20);
    }
    if (i < H) {
        float fy = (float)(((double)i + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int b0 = cv_round_pos((1.0f - fy) * 2048.0f), b1 = cv_round_pos(fy * 2048.0f);
        const int sy0 = min(max(sy, 0), ch - 1), sy1 = min(max(sy + 1, 0), ch - 1);
        ytab[i].ofs = sy0 | (sy1 << 16);
        ytab[i].w = (uint32_t)b0 | ((uint32_t)b1 << 16);
    }
}

// B | G << 8 | R << 16 of the pixel at byte offset o; the 4-byte load of the very last pixel of the stack is
// shifted back by one byte instead of running past the allocation.
__device__ __forceinline__ uint32_t load_bgr(const uint8_t* __restrict__ frame, uint32_t o, size_t limit)
{
    uint32_t v;
    if ((size_t)o + 4 <= limit) {
        __builtin_memcpy(&v, frame + o, 4);
    } else {
        __builtin_memcpy(&v, frame + o - 1, 4);
        v >>= 8;
    }
    return v & 0xFFFFFFu;
}

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b), c, false);
}

constexpr int kRowPitch = 800;        // bytes of one staged source row in LDS: 50 chunks of 16 bytes (256 output px + slack)

// Workgroup = 8 output rows x 256 pixels; wavefront = kRows consecutive rows; lane = 4 consecutive pixels per row (one
// 12-byte store each).
// The two source rows of an output row are shared by all its pixels: the wavefront copies the span it needs of both
// (<= 800 bytes each, from the dword holding the first tap) into LDS with two global->LDS 16-byte loads per lane and
// takes the taps from there (three dword reads per pixel and row + v_alignbyte for the 3-byte-pixel misalignment).
// That replaces eight unaligned 8-byte loads per lane, whose instruction count -- not bytes -- bounded the first version.
// High 32 bits of the product of two 24-bit values (v_mul_hi_u32_u24).
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b)
{
    return (uint32_t)(((unsigned long long)(a & 0xFFFFFFu) * (unsigned long long)(b & 0xFFFFFFu)) >> 32);
}

constexpr int kRows = 2;              // output rows per wavefront (amortises the scalar set-up and the column table); 1: +8 %, 4: +4 %
constexpr int kWaves = 4;             // wavefronts per workgroup (they never cooperate; 1, 2 and 4 measure the same)

__global__ __launch_bounds__(64 * kWaves) void resize_kernel(const uint8_t* __restrict__ frames, uint8_t* __restrict__ out, int n,
                                                     int W, int H, int left, int top, int cw,
                                                     const ResizeTab* __restrict__ xtab,
                                                     const ResizeTab* __restrict__ ytab, TileOrder order)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[kWaves][kRows][2 * kRowPitch + 64];
    int f, tile_y, tile_x;
    if (!order.decode(blockIdx.x, f, tile_y, tile_x)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ya = (tile_y * kWaves + wave) * kRows;
    const int xw = tile_x * 256, x0 = xw + lane * 4;
    if (ya >= H) return;
    const size_t frame_bytes = (size_t)W * H * 3;
    const uint8_t* __restrict__ src = frames + (size_t)f * frame_bytes;
    uint8_t* __restrict__ dst = out + (size_t)f * frame_bytes;
    const size_t limit = (size_t)(n - f) * frame_bytes;
    const size_t base = (size_t)(uintptr_t)src;

    // span of source columns this wavefront touches: taps sx .. sx+1 for its first .. last pixel (the tables are
    // monotone); a row is staged when the span fits the LDS row and the 16-byte chunks stay inside the frame stack
    const uint32_t sx_first = (uint32_t)xtab[xw].ofs, sx_last = (uint32_t)xtab[min(xw + 255, W - 1)].ofs;
    const uint32_t span = 3u * (sx_last + 2u - sx_first);
    uint32_t row0[kRows], row1[kRows], b0s[kRows], b1s[kRows], s0[kRows], s1[kRows];
    bool staged[kRows];
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const ResizeTab yt = ytab[min(ya + q, H - 1)];
        b0s[q] = (yt.w & 0xFFFFu) << 8;
        b1s[q] = (yt.w >> 16) << 8;
        row0[q] = (uint32_t)(top + (yt.ofs & 0xFFFF)) * (uint32_t)W + (uint32_t)left;
        row1[q] = (uint32_t)(top + (yt.ofs >> 16)) * (uint32_t)W + (uint32_t)left;
        const size_t g0 = (size_t)(row0[q] + sx_first) * 3u, g1 = (size_t)(row1[q] + sx_first) * 3u;     // byte offsets in the frame
        s0[q] = (uint32_t)((base + g0) & 3u);                                                          // misalignment of each row
        s1[q] = (uint32_t)((base + g1) & 3u);
        staged[q] = ya + q < H && span + 3u + 12u <= (uint32_t)kRowPitch && g0 >= 3u && g1 >= 3u &&
                    (g0 > g1 ? g0 : g1) - 3u + (size_t)kRowPitch <= limit;
        if (staged[q] && lane < kRowPitch / 16) {
            uint32_t o = (uint32_t)lane << 4;
            asm("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (g0 - s0[q]) + o),
                                             (__attribute__((address_space(3))) void*)&s_rows[wave][q][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (g1 - s1[q]) + o),
                                             (__attribute__((address_space(3))) void*)&s_rows[wave][q][kRowPitch], 16, 0, 0);
        }
    }
    ResizeTab xt[4];
    if (x0 < W) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xt[j] = xtab[min(x0 + j, W - 1)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // staged rows (and the column table) have landed
    if (x0 >= W) return;

#pragma unroll
    for (int q = 0; q < kRows; ++q) {
        const int y = ya + q;
        if (y >= H) break;
        uint32_t px[4];
        // Fast path: the lane's four pixels are inside the frame and every tap load stays inside the frame stack.
        // Where sx is the last column of the crop the second weight is 0, so whatever lies right of it may be read.
        const bool whole = x0 + 3 < W && ((size_t)(max(row0[q], row1[q]) + (uint32_t)cw) * 3u + 8u <= limit);
        if (staged[q] || whole) {
            // per pixel and channel (S[sx], S[sx+1]) as two uint16 in one register -- {B, G, R} of source row 0, then of row 1
            uint32_t fld[4][6];
            if (staged[q]) {
                // from the staged rows by byte loads with immediate offsets: ds_read_u8 puts S[sx] into the low byte of one register,
                // ds_read_u8_d16_hi S[sx+1] into bits 16-23 of another (with SRAM ECC a d16 load zeroes the other half of its
                // destination), one v_or_b32 joins them -- instead of three dword loads, two v_alignbyte_b32 and three v_perm_b32
                // per pixel and row (the same scheme as the warp kernel's taps, warp.hip)
                uint32_t lo[4][6], hi[4][6];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t rel = 3u * ((uint32_t)xt[j].ofs - sx_first);
                    const uint32_t at0 = (uint32_t)(uintptr_t)&s_rows[wave][q][0] + rel + s0[q];
                    const uint32_t at1 = (uint32_t)(uintptr_t)&s_rows[wave][q][kRowPitch] + rel + s1[q];
                    asm volatile("ds_read_u8 %0, %6 offset:0\n\tds_read_u8_d16_hi %1, %6 offset:3\n\t"
                                 "ds_read_u8 %2, %6 offset:1\n\tds_read_u8_d16_hi %3, %6 offset:4\n\t"
                                 "ds_read_u8 %4, %6 offset:2\n\tds_read_u8_d16_hi %5, %6 offset:5"
                                 : "=&v"(lo[j][0]), "=&v"(hi[j][0]), "=&v"(lo[j][1]), "=&v"(hi[j][1]), "=&v"(lo[j][2]), "=&v"(hi[j][2])
                                 : "v"(at0));
                    asm volatile("ds_read_u8 %0, %6 offset:0\n\tds_read_u8_d16_hi %1, %6 offset:3\n\t"
                                 "ds_read_u8 %2, %6 offset:1\n\tds_read_u8_d16_hi %3, %6 offset:4\n\t"
                                 "ds_read_u8 %4, %6 offset:2\n\tds_read_u8_d16_hi %5, %6 offset:5"
                                 : "=&v"(lo[j][3]), "=&v"(hi[j][3]), "=&v"(lo[j][4]), "=&v"(hi[j][4]), "=&v"(lo[j][5]), "=&v"(hi[j][5])
                                 : "v"(at1));
                    if (j & 1) {
                        // the compiler does not see these loads: wait here (two pixels' worth in flight) and hand the registers over
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(lo[j - 1][0]), "+v"(hi[j - 1][0]), "+v"(lo[j - 1][1]), "+v"(hi[j - 1][1]), "+v"(lo[j - 1][2]), "+v"(hi[j - 1][2]),
                                       "+v"(lo[j - 1][3]), "+v"(hi[j - 1][3]), "+v"(lo[j - 1][4]), "+v"(hi[j - 1][4]), "+v"(lo[j - 1][5]), "+v"(hi[j - 1][5]),
                                       "+v"(lo[j][0]), "+v"(hi[j][0]), "+v"(lo[j][1]), "+v"(hi[j][1]), "+v"(lo[j][2]), "+v"(hi[j][2]),
                                       "+v"(lo[j][3]), "+v"(hi[j][3]), "+v"(lo[j][4]), "+v"(hi[j][4]), "+v"(lo[j][5]), "+v"(hi[j][5]) :: "memory");
#pragma unroll
                        for (int c = 0; c < 6; ++c) { fld[j - 1][c] = lo[j - 1][c] | hi[j - 1][c]; fld[j][c] = lo[j][c] | hi[j][c]; }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint2 a, b;                                              // bytes: B0 G0 R0 B1 | G1 R1 . .
                    __builtin_memcpy(&a, src + (row0[q] + (uint32_t)xt[j].ofs) * 3u, 8);
                    __builtin_memcpy(&b, src + (row1[q] + (uint32_t)xt[j].ofs) * 3u, 8);
                    fld[j][0] = __builtin_amdgcn_perm(a.y, a.x, 0x0C030C00u); fld[j][3] = __builtin_amdgcn_perm(b.y, b.x, 0x0C030C00u);
                    fld[j][1] = __builtin_amdgcn_perm(a.y, a.x, 0x0C040C01u); fld[j][4] = __builtin_amdgcn_perm(b.y, b.x, 0x0C040C01u);
                    fld[j][2] = __builtin_amdgcn_perm(a.y, a.x, 0x0C050C02u); fld[j][5] = __builtin_amdgcn_perm(b.y, b.x, 0x0C050C02u);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // v_dot2_u32_u16 with the weights pre-scaled by 16: T = 16 t < 2^24, T & ~255 = 256 (t >> 4), and
                // (b * (t >> 4)) >> 16 is the high half of the 24 x 24-bit product (256 b) * (256 (t >> 4)): one v_and + one
                // v_mul_hi_u32_u24 per term
                const uint32_t w = xt[j].w;                                        // 16 a0 | 16 a1 << 16
                const uint32_t tB0 = udot2(fld[j][0], w, 0u), tG0 = udot2(fld[j][1], w, 0u), tR0 = udot2(fld[j][2], w, 0u);
                const uint32_t tB1 = udot2(fld[j][3], w, 0u), tG1 = udot2(fld[j][4], w, 0u), tR1 = udot2(fld[j][5], w, 0u);
                const uint32_t vB = (mulhi_u24(b0s[q], tB0 & ~255u) + mulhi_u24(b1s[q], tB1 & ~255u) + 2u) >> 2;
                const uint32_t vG = (mulhi_u24(b0s[q], tG0 & ~255u) + mulhi_u24(b1s[q], tG1 & ~255u) + 2u) >> 2;
                const uint32_t vR = (mulhi_u24(b0s[q], tR0 & ~255u) + mulhi_u24(b1s[q], tR1 & ~255u) + 2u) >> 2;
                // no saturation needed: each weight pair sums to 2048 +- 1 (two cvRound of complementary fractions), so
                // t <= 255 * 2049, t >> 4 <= 32655 and the two high halves sum to at most 2049 * 32655 / 65536 < 1021,
                // i.e. (sum + 2) >> 2 <= 255 -- cv2's saturate_cast never triggers either
                px[j] = vB | (vG << 8) | (vR << 16);
            }
        } else {
            const uint32_t b0 = b0s[q] >> 8, b1 = b1s[q] >> 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                px[j] = 0;
                if (x0 + j >= W) continue;
                const uint32_t a0 = (xt[j].w & 0xFFFFu) >> 4, a1 = xt[j].w >> 20;
                const uint32_t sx = (uint32_t)xt[j].ofs, sx1 = min(sx + 1u, (uint32_t)(cw - 1));     // a1 == 0 where sx == cw-1
                const uint32_t p00 = load_bgr(src, (row0[q] + sx) * 3u, limit), p01 = load_bgr(src, (row0[q] + sx1) * 3u, limit);
                const uint32_t p10 = load_bgr(src, (row1[q] + sx) * 3u, limit), p11 = load_bgr(src, (row1[q] + sx1) * 3u, limit);
                uint32_t r = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t t0 = ((p00 >> (8 * c)) & 255u) * a0 + ((p01 >> (8 * c)) & 255u) * a1;
                    const uint32_t t1 = ((p10 >> (8 * c)) & 255u) * a0 + ((p11 >> (8 * c)) & 255u) * a1;
                    const uint32_t v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2u) >> 2;
                    r |= min(v, 255u) << (8 * c);
                }
                px[j] = r;
            }
        }
        const uint32_t o = ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u;
        if ((W & 3) == 0 && x0 + 3 < W) {
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

size_t crop_resize_workspace_bytes(int W, int H) { return (size_t)(W + H) * sizeof(ResizeTab); }

int launch_crop_resize(const uint8_t* frames, uint8_t* out, int n, int W, int H, int left, int top, int right, int bottom,
                       void* work, hipStream_t st)
{
    if (const int rc = check_d16_zero_fill(st)) return rc;
    if (n <= 0 || n > 65535 || W < 1 || H < 1 || W > 32767 || H > 32767) {
        set_error("mf_crop_resize_u8c3: unsupported shape n=%d W=%d H=%d", n, W, H);
        return MF_ERR_INVALID_ARG;
    }
    if (left < 0 || top < 0 || right >= W || bottom >= H || right < left || bottom < top) {
        set_error("mf_crop_resize_u8c3: empty or out-of-frame crop rectangle (%d, %d, %d, %d) for %dx%d (cv2.resize would "
                  "fail on an empty source)", left, top, right, bottom, W, H);
        return MF_ERR_INVALID_ARG;
    }
    const int cw = right - left + 1, ch = bottom - top + 1;
    const double scale_x = 1.0 / ((double)W / (double)cw), scale_y = 1.0 / ((double)H / (double)ch);
    ResizeTab* xtab = (ResizeTab*)work;
    ResizeTab* ytab = xtab + W;
    const int m = W > H ? W : H;
    hipLaunchKernelGGL(resize_tables_kernel, dim3((m + 255) / 256), dim3(256), 0, st, cw, ch, W, H, scale_x, scale_y, xtab, ytab);
    int rc = hip_fail(hipGetLastError(), "resize_tables_kernel launch");
    if (rc != MF_OK) return rc;
    TileOrder order;
    if (!make_tile_order((W + 255) / 256, (H + kWaves * kRows - 1) / (kWaves * kRows), n, order)) {
        set_error("mf_crop_resize_u8c3: too many tiles");
        return MF_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(resize_kernel, dim3(order.per_xcd * 8u), dim3(64 * kWaves), 0, st, frames, out, n, W, H, left, top, cw, xtab, ytab,
                       order);
    return hip_fail(hipGetLastError(), "resize_kernel launch");
}

}  // namespace mf

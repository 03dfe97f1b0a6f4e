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
    } else if (o != 0) {
        __builtin_memcpy(&v, frame + o - 1, 4);
        v >>= 8;
    } else {                                             // a stack of ONE pixel: nothing in front of it either
        v = (uint32_t)frame[0] | (uint32_t)frame[1] << 8 | (uint32_t)frame[2] << 16;
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

constexpr int kRows = 8;              // output rows per wavefront
constexpr int kWaves = 4;             // wavefronts per workgroup (they never cooperate)
constexpr int kSrcRows = kRows + 1;   // source rows a wavefront stages: _crop_frames only ever scales UP (the crop lies inside the frame), so
                                      // consecutive output rows advance by at most one source row

// The horizontal pass of ONE staged source row for the lane's four pixels: t = S[sx] a0 + S[sx+1] a1 per channel (v_dot2_u32_u16 with
// the weights pre-scaled by 16: T = 16 t < 2^24), returned as T & ~255 = 256 (t >> 4), what the vertical pass multiplies.  The taps
// are byte loads with immediate offsets: ds_read_u8 puts S[sx] into the low byte of one register, ds_read_u8_d16_hi S[sx+1] into
// bits 16-23 of another (with SRAM ECC a d16 load zeroes the other half of its destination: check_d16_zero_fill), one v_or_b32 joins
// them.  All 24 loads and their wait sit in ONE asm block: nothing can be scheduled between issue and wait.
__device__ __forceinline__ void hpass_row(const uint32_t (&at)[4], const uint32_t (&w)[4], uint32_t (&T)[4][3])
{
    uint32_t lo[4][3], hi[4][3];
    asm volatile("ds_read_u8 %0, %24 offset:0\n\tds_read_u8_d16_hi %1, %24 offset:3\n\t"
                 "ds_read_u8 %2, %24 offset:1\n\tds_read_u8_d16_hi %3, %24 offset:4\n\t"
                 "ds_read_u8 %4, %24 offset:2\n\tds_read_u8_d16_hi %5, %24 offset:5\n\t"
                 "ds_read_u8 %6, %25 offset:0\n\tds_read_u8_d16_hi %7, %25 offset:3\n\t"
                 "ds_read_u8 %8, %25 offset:1\n\tds_read_u8_d16_hi %9, %25 offset:4\n\t"
                 "ds_read_u8 %10, %25 offset:2\n\tds_read_u8_d16_hi %11, %25 offset:5\n\t"
                 "ds_read_u8 %12, %26 offset:0\n\tds_read_u8_d16_hi %13, %26 offset:3\n\t"
                 "ds_read_u8 %14, %26 offset:1\n\tds_read_u8_d16_hi %15, %26 offset:4\n\t"
                 "ds_read_u8 %16, %26 offset:2\n\tds_read_u8_d16_hi %17, %26 offset:5\n\t"
                 "ds_read_u8 %18, %27 offset:0\n\tds_read_u8_d16_hi %19, %27 offset:3\n\t"
                 "ds_read_u8 %20, %27 offset:1\n\tds_read_u8_d16_hi %21, %27 offset:4\n\t"
                 "ds_read_u8 %22, %27 offset:2\n\tds_read_u8_d16_hi %23, %27 offset:5\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo[0][0]), "=&v"(hi[0][0]), "=&v"(lo[0][1]), "=&v"(hi[0][1]), "=&v"(lo[0][2]), "=&v"(hi[0][2]),
                   "=&v"(lo[1][0]), "=&v"(hi[1][0]), "=&v"(lo[1][1]), "=&v"(hi[1][1]), "=&v"(lo[1][2]), "=&v"(hi[1][2]),
                   "=&v"(lo[2][0]), "=&v"(hi[2][0]), "=&v"(lo[2][1]), "=&v"(hi[2][1]), "=&v"(lo[2][2]), "=&v"(hi[2][2]),
                   "=&v"(lo[3][0]), "=&v"(hi[3][0]), "=&v"(lo[3][1]), "=&v"(hi[3][1]), "=&v"(lo[3][2]), "=&v"(hi[3][2])
                 : "v"(at[0]), "v"(at[1]), "v"(at[2]), "v"(at[3]) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[j][c] = udot2(lo[j][c] | hi[j][c], w[j], 0u) & ~255u;
}

// The vertical pass + store of one output row: out = (((b0 (t0 >> 4)) >> 16) + ((b1 (t1 >> 4)) >> 16) + 2) >> 2, each product's
// high half by one v_mul_hi_u32_u24 of (256 b) and (256 (t >> 4)).  No saturation needed: each weight pair sums to 2048 +- 1 (two
// cvRound of complementary fractions), so t <= 255 * 2049, t >> 4 <= 32655 and the two high halves sum to at most
// 2049 * 32655 / 65536 < 1021, i.e. (sum + 2) >> 2 <= 255 -- cv2's saturate_cast never triggers either.
__device__ __forceinline__ void vpass_store(const uint32_t (&T0)[4][3], const uint32_t (&T1)[4][3], uint32_t b0s, uint32_t b1s,
                                            uint8_t* __restrict__ dst, uint32_t o, int x0, int W)
{
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t vB = (mulhi_u24(b0s, T0[j][0]) + mulhi_u24(b1s, T1[j][0]) + 2u) >> 2;
        const uint32_t vG = (mulhi_u24(b0s, T0[j][1]) + mulhi_u24(b1s, T1[j][1]) + 2u) >> 2;
        const uint32_t vR = (mulhi_u24(b0s, T0[j][2]) + mulhi_u24(b1s, T1[j][2]) + 2u) >> 2;
        px[j] = vB | (vG << 8) | (vR << 16);
    }
    if (x0 + 3 < W) {
        uint3 d;
        d.x = px[0] | (px[1] << 24);
        d.y = (px[1] >> 8) | (px[2] << 16);
        d.z = (px[2] >> 16) | (px[3] << 8);
        if ((W & 3) == 0) *reinterpret_cast<uint3*>(dst + o) = d;
        else __builtin_memcpy(dst + o, &d, 12);                  // (a row of W % 4 != 0 starts anywhere: one unaligned 12-byte store)
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

// Workgroup = kWaves wavefronts; wavefront = kRows consecutive output rows x 256 pixels; lane = 4 consecutive pixels per row (one
// 12-byte store each).  The source rows of an output row are shared by all its pixels, and -- the crop only ever scales UP -- by the
// NEXT output row too: the kRows output rows of a wavefront read at most kRows + 1 source rows.  The wavefront copies the span it
// needs of each (<= 800 bytes, from the dword holding the first tap) into LDS with ONE global->LDS 16-byte load per row, runs the
// horizontal pass once per SOURCE row (the result of an output row's second source row is the next output row's first) and the
// vertical pass per output row: (kRows + 1) / kRows horizontal passes per output row instead of 2, 9 window copies per 8 rows
// instead of 16.  Anything that cannot be staged (a frame narrower than a chunk, the last rows of the stack, a call that scales down)
// takes the direct path below, row by row.
__global__ __launch_bounds__(64 * kWaves) void resize_kernel(const uint8_t* __restrict__ frames, uint8_t* __restrict__ out, int n,
                                                     int W, int H, int left, int top, int cw,
                                                     const ResizeTab* __restrict__ xtab,
                                                     const ResizeTab* __restrict__ ytab, TileOrder order)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[kWaves][kSrcRows][kRowPitch + 16];
    int f, tile_y, tile_x;
    if (!order.decode(blockIdx.x, f, tile_y, tile_x)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ya = (tile_y * kWaves + wave) * kRows;
    const int xw = tile_x * 256, x0 = xw + lane * 4;
    if (ya >= H) return;
    const int rows = min(kRows, H - ya);
    const size_t frame_bytes = (size_t)W * H * 3;
    const uint8_t* __restrict__ src = frames + (size_t)f * frame_bytes;
    uint8_t* __restrict__ dst = out + (size_t)f * frame_bytes;
    const size_t limit = (size_t)(n - f) * frame_bytes;
    const size_t base = (size_t)(uintptr_t)src;

    // span of source columns this wavefront touches: taps sx .. sx+1 for its first .. last pixel (the tables are monotone), and of
    // source rows: sy0 of its first .. sy1 of its last output row
    const uint32_t sx_first = (uint32_t)xtab[xw].ofs, sx_last = (uint32_t)xtab[min(xw + 255, W - 1)].ofs;
    const uint32_t span = 3u * (sx_last + 2u - sx_first);
    const int r_first = ytab[ya].ofs & 0xFFFF, r_last = ytab[ya + rows - 1].ofs >> 16;
    const int nsrc = r_last - r_first + 1;
    // byte offset in the frame of the first tap of source row i: g(i) = ((top + r_first + i) W + left + sx_first) 3
    const size_t g_first = ((size_t)(top + r_first) * (size_t)W + (size_t)left + sx_first) * 3u;
    const size_t g_last = g_first + (size_t)(nsrc - 1) * (size_t)W * 3u;
    const bool staged = nsrc <= kSrcRows && span + 3u + 12u <= (uint32_t)kRowPitch && g_first >= 3u && g_last - 3u + (size_t)kRowPitch <= limit;
    if (staged && lane < kRowPitch / 16) {
        uint32_t o = (uint32_t)lane << 4;
        asm("" : "+v"(o));
#pragma unroll 1
        for (int i = 0; i < nsrc; ++i) {
            const size_t g = g_first + (size_t)i * (size_t)W * 3u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (g - ((base + g) & 3u)) + o),
                                             (__attribute__((address_space(3))) void*)&s_rows[wave][i][0], 16, 0, 0);
        }
    }
    ResizeTab xt[4];
    if (x0 < W) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xt[j] = xtab[min(x0 + j, W - 1)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // staged rows (and the column table) have landed
    if (x0 >= W) return;

    if (staged) {
        uint32_t rel[4], wq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { rel[j] = 3u * ((uint32_t)xt[j].ofs - sx_first) + (uint32_t)(uintptr_t)&s_rows[wave][0][0]; wq[j] = xt[j].w; }
        const uint32_t mis0 = (uint32_t)((base + g_first) & 3u), mis_step = (3u * (uint32_t)W) & 3u;      // misalignment of row i: (mis0 + i mis_step) & 3
        const auto row_at = [&](int i, uint32_t (&at)[4]) {
            const uint32_t add = (uint32_t)i * (uint32_t)(kRowPitch + 16) + ((mis0 + (uint32_t)i * mis_step) & 3u);
#pragma unroll
            for (int j = 0; j < 4; ++j) at[j] = rel[j] + add;
        };
        // Two register sets take turns as "first source row" and "second source row" of an output row (no copies): an output row
        // whose first source row is the previous one's second reuses its horizontal pass.
        uint32_t Ta[4][3], Tb[4][3], at[4];
        int have_a = -1, have_b = -1;                                  // source row (relative) each set holds
#pragma unroll 1
        for (int q = 0; q < rows; q += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int y = ya + q + h;
                if (y >= ya + rows) break;
                const ResizeTab yt = ytab[y];
                const int i0 = (yt.ofs & 0xFFFF) - r_first, i1 = (yt.ofs >> 16) - r_first;
                const uint32_t b0s = (yt.w & 0xFFFFu) << 8, b1s = (yt.w >> 16) << 8;
                const uint32_t o = ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u;
                if (h == 0) {                                          // first source row in set A, second in set B
                    if (have_a != i0) { row_at(i0, at); hpass_row(at, wq, Ta); have_a = i0; }
                    if (have_b != i1) { row_at(i1, at); hpass_row(at, wq, Tb); have_b = i1; }
                    vpass_store(Ta, Tb, b0s, b1s, dst, o, x0, W);
                } else {                                               // ... and the other way round: this row's first is usually set B
                    if (have_b != i0) { row_at(i0, at); hpass_row(at, wq, Tb); have_b = i0; }
                    if (have_a != i1) { row_at(i1, at); hpass_row(at, wq, Ta); have_a = i1; }
                    vpass_store(Tb, Ta, b0s, b1s, dst, o, x0, W);
                }
            }
        }
        return;
    }

    // direct path: taps straight from the frame, row by row
#pragma unroll 1
    for (int q = 0; q < rows; ++q) {
        const int y = ya + q;
        const ResizeTab yt = ytab[y];
        const uint32_t b0s = (yt.w & 0xFFFFu) << 8, b1s = (yt.w >> 16) << 8;
        const uint32_t row0 = (uint32_t)(top + (yt.ofs & 0xFFFF)) * (uint32_t)W + (uint32_t)left;
        const uint32_t row1 = (uint32_t)(top + (yt.ofs >> 16)) * (uint32_t)W + (uint32_t)left;
        const uint32_t o = ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u;
        // Fast form: the lane's four pixels are inside the frame and every tap load stays inside the frame stack.
        // Where sx is the last column of the crop the second weight is 0, so whatever lies right of it may be read.
        const bool whole = x0 + 3 < W && ((size_t)(max(row0, row1) + (uint32_t)cw) * 3u + 8u <= limit);
        if (whole) {
            uint32_t T0[4][3], T1[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 a, b;                                              // bytes: B0 G0 R0 B1 | G1 R1 . .
                __builtin_memcpy(&a, src + (row0 + (uint32_t)xt[j].ofs) * 3u, 8);
                __builtin_memcpy(&b, src + (row1 + (uint32_t)xt[j].ofs) * 3u, 8);
                const uint32_t w = xt[j].w;
                T0[j][0] = udot2(__builtin_amdgcn_perm(a.y, a.x, 0x0C030C00u), w, 0u) & ~255u; T1[j][0] = udot2(__builtin_amdgcn_perm(b.y, b.x, 0x0C030C00u), w, 0u) & ~255u;
                T0[j][1] = udot2(__builtin_amdgcn_perm(a.y, a.x, 0x0C040C01u), w, 0u) & ~255u; T1[j][1] = udot2(__builtin_amdgcn_perm(b.y, b.x, 0x0C040C01u), w, 0u) & ~255u;
                T0[j][2] = udot2(__builtin_amdgcn_perm(a.y, a.x, 0x0C050C02u), w, 0u) & ~255u; T1[j][2] = udot2(__builtin_amdgcn_perm(b.y, b.x, 0x0C050C02u), w, 0u) & ~255u;
            }
            vpass_store(T0, T1, b0s, b1s, dst, o, x0, W);
            continue;
        }
        const uint32_t b0 = b0s >> 8, b1 = b1s >> 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (x0 + j >= W) continue;
            const uint32_t a0 = (xt[j].w & 0xFFFFu) >> 4, a1 = xt[j].w >> 20;
            const uint32_t sx = (uint32_t)xt[j].ofs, sx1 = min(sx + 1u, (uint32_t)(cw - 1));     // a1 == 0 where sx == cw-1
            const uint32_t p00 = load_bgr(src, (row0 + sx) * 3u, limit), p01 = load_bgr(src, (row0 + sx1) * 3u, limit);
            const uint32_t p10 = load_bgr(src, (row1 + sx) * 3u, limit), p11 = load_bgr(src, (row1 + sx1) * 3u, limit);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t t0 = ((p00 >> (8 * c)) & 255u) * a0 + ((p01 >> (8 * c)) & 255u) * a1;
                const uint32_t t1 = ((p10 >> (8 * c)) & 255u) * a0 + ((p11 >> (8 * c)) & 255u) * a1;
                const uint32_t v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2u) >> 2;
                dst[o + 3 * j + c] = (uint8_t)min(v, 255u);
            }
        }
    }
}

size_t crop_resize_workspace_bytes(int W, int H) { return (size_t)(W + H) * sizeof(ResizeTab); }

int launch_crop_resize(const uint8_t* frames, uint8_t* out, int n, int W, int H, int left, int top, int right, int bottom,
                       void* work, hipStream_t st)
{
    if (const int rc = check_d16_zero_fill(st)) return rc;
    if (n <= 0 || W < 1 || H < 1 || W > 32767 || H > 32767) {           // (any number of frames that make_tile_order can count: 2^31 tiles)
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

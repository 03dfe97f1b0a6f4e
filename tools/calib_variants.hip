// Which part of the warp kernel's memory pattern is slow?  Same tiling as warp_kernel (128x16 tiles, lane = 4 px).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H)
{
    const int f = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 128 + wave * 32 + (lane & 7) * 4;
    const uint8_t* src = in + (size_t)f * W * H * 3;
    uint8_t* dst = out + (size_t)f * W * H * 3;
    for (int q = 0; q < 2; ++q) {
        const int y = blockIdx.y * 16 + q * 8 + (lane >> 3);
        if (y >= H - 1 || x0 + 6 >= W) continue;
        uint32_t acc[4] = {0, 0, 0, 0};
        if (MODE == 0 || MODE == 1) {           // two unaligned 8-byte loads per pixel
            for (int j = 0; j < 4; ++j) {
                const uint32_t o = ((uint32_t)y * W + x0 + j) * 3u;
                uint2 a, b; __builtin_memcpy(&a, src + o, 8); __builtin_memcpy(&b, src + o + 3u * W, 8);
                acc[j] = (a.x ^ a.y ^ b.x ^ b.y) & 0xFFFFFFu;
            }
        } else if (MODE == 2) {                 // one aligned 16-byte load per row (covers the 4 px + 1): 2 loads per lane
            const uint32_t o = ((uint32_t)y * W + x0) * 3u;       // 12-byte aligned -> 4-byte aligned
            uint4 a, b; __builtin_memcpy(&a, src + o, 16); __builtin_memcpy(&b, src + o + 3u * W, 16);
            acc[0] = a.x ^ b.x; acc[1] = a.y ^ b.y; acc[2] = a.z ^ b.z; acc[3] = (a.w ^ b.w) & 0xFFFFFF;
        } else if (MODE == 3) {                 // no loads
            acc[0] = x0; acc[1] = y; acc[2] = f; acc[3] = lane;
        } else if (MODE == 4) {                 // four unaligned 4-byte loads per pixel
            for (int j = 0; j < 4; ++j) {
                const uint32_t o = ((uint32_t)y * W + x0 + j) * 3u;
                uint32_t a, b, c, d; __builtin_memcpy(&a, src + o, 4); __builtin_memcpy(&b, src + o + 3, 4);
                __builtin_memcpy(&c, src + o + 3u * W, 4); __builtin_memcpy(&d, src + o + 3u * W + 3, 4);
                acc[j] = (a ^ b ^ c ^ d) & 0xFFFFFFu;
            }
        }
        if (MODE == 7) {                        // LDS-staged via global_load_lds_dwordx4 (no VGPR staging, no ds_write)
            __shared__ __attribute__((aligned(16))) uint8_t s_dma[4][12 * 160 + 128];
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const int xa = blockIdx.x * 128 + wv * 32, ya = blockIdx.y * 16 + q * 8;
            const size_t row_bytes = (size_t)W * 3;
            const uint32_t start = (uint32_t)xa * 3u;
            const uint32_t bs = start & ~3u, shift = start & 3u;                  // dword-aligned region start in the row
            const uint8_t* gbase = src + (size_t)ya * row_bytes + bs;
            uint8_t* lds = s_dma[wv];
            const int r0 = (lane * 205) >> 11, c0 = lane - 10 * r0;                // lane / 10 for lane < 128
            const int l1 = lane + 64, r1 = (l1 * 205) >> 11, c1 = l1 - 10 * r1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + (size_t)r0 * row_bytes + 16 * c0),
                                             (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
            if (lane < 56)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + (size_t)r1 * row_bytes + 16 * c1),
                                                 (__attribute__((address_space(3))) void*)(lds + 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ry = (lane >> 3);
            for (int j = 0; j < 4; ++j) {
                const uint32_t a = (uint32_t)ry * 160u + 3u * (uint32_t)((lane & 7) * 4 + j) + shift;
                const uint32_t* p = reinterpret_cast<const uint32_t*>(lds + (a & ~3u));
                const uint32_t t0 = p[0], t1 = p[1], t2 = p[2], b0 = p[40], b1 = p[41], b2 = p[42];
                const uint32_t ax = __builtin_amdgcn_alignbyte(t1, t0, a), ay = __builtin_amdgcn_alignbyte(t2, t1, a);
                const uint32_t bx = __builtin_amdgcn_alignbyte(b1, b0, a), by = __builtin_amdgcn_alignbyte(b2, b1, a);
                acc[j] = (ax ^ ay ^ bx ^ by) & 0xFFFFFFu;
            }
        }
        if (MODE == 5 || MODE == 6) {           // LDS-staged source rows: 2 aligned 16-byte loads per lane, taps from LDS
            __shared__ __attribute__((aligned(16))) uint8_t s_rows[4][10 * 160];
            const int xa = blockIdx.x * 128 + wave * 32, ya = blockIdx.y * 16 + q * 8;
            const size_t row_bytes = (size_t)W * 3;
            const size_t base = ((size_t)ya * row_bytes + (size_t)xa * 3) & ~(size_t)15;   // 16-byte aligned start of the region
            const uint32_t shift = (uint32_t)(((size_t)ya * row_bytes + (size_t)xa * 3) - base);
            uint8_t* lds = s_rows[wave];
            for (int c = lane; c < 100; c += 64) {          // 10 rows x 10 chunks of 16 bytes
                const int rr = c / 10, cc = c - rr * 10;
                // rows are row_bytes apart; keep 16-byte alignment per row by re-aligning each row start
                const size_t ro = (base + (size_t)rr * row_bytes) & ~(size_t)15;
                const uint4 v = *reinterpret_cast<const uint4*>(src + ro + 16 * cc);
                *reinterpret_cast<uint4*>(lds + rr * 160 + 16 * cc) = v;
            }
            // (single wave owns its region: LDS ops of a wave complete in order)
            const int ry = (lane >> 3);
            for (int j = 0; j < 4; ++j) {
                const uint32_t a = (uint32_t)ry * 160u + 3u * (uint32_t)((lane & 7) * 4 + j) + (shift & 15u);
                const uint32_t a4 = a & ~3u;
                uint32_t t0, t1, t2, b0, b1, b2;
                if (MODE == 5) {
                    const uint3 t = *reinterpret_cast<const uint3*>(lds + a4);
                    const uint3 b = *reinterpret_cast<const uint3*>(lds + a4 + 160);
                    t0 = t.x; t1 = t.y; t2 = t.z; b0 = b.x; b1 = b.y; b2 = b.z;
                } else {
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(lds + a4);
                    t0 = p[0]; t1 = p[1]; t2 = p[2]; b0 = p[40]; b1 = p[41]; b2 = p[42];
                }
                const uint32_t ax = __builtin_amdgcn_alignbyte(t1, t0, a), ay = __builtin_amdgcn_alignbyte(t2, t1, a);
                const uint32_t bx = __builtin_amdgcn_alignbyte(b1, b0, a), by = __builtin_amdgcn_alignbyte(b2, b1, a);
                acc[j] = (ax ^ ay ^ bx ^ by) & 0xFFFFFFu;
            }
        }
        if (MODE == 1) {                        // loads only: store one byte per wave-row rarely
            if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) dst[0] = 1;
        } else {
            uint3 d; d.x = acc[0] | (acc[1] << 24); d.y = (acc[1] >> 8) | (acc[2] << 16); d.z = (acc[2] >> 16) | (acc[3] << 8);
            *reinterpret_cast<uint3*>(dst + ((size_t)y * W + x0) * 3) = d;
        }
    }
}
template <int MODE> void run(const char* name, const uint8_t* in, uint8_t* out)
{
    const int W = 1920, H = 1080, n = 300;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<dim3(15, 68, n), 256>>>(in, out, W, H);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<dim3(15, 68, n), 256>>>(in, out, W, H);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %.3f ms\n", name, ms / 5);
}
int main()
{
    const size_t N = (size_t)1920 * 1080 * 3 * 300;
    uint8_t *in, *out; hipMalloc(&in, N + 64); hipMalloc(&out, N + 64);
    hipMemset(in, 0x5A, N + 64); hipMemset(out, 0, N + 64);
    run<0>("2x unaligned 8B loads/px + 12B store", in, out);
    run<1>("2x unaligned 8B loads/px, no store", in, out);
    run<2>("2x 16B loads per 4 px + 12B store", in, out);
    run<3>("no loads, 12B store", in, out);
    run<4>("4x unaligned 4B loads/px + 12B store", in, out);
    run<5>("LDS-staged rows, ds_read 12B x2/px + 12B store", in, out);
    run<6>("LDS-staged rows, 3 dword reads x2/px + 12B store", in, out);
    run<7>("LDS-DMA staged rows (global_load_lds x4) + 12B store", in, out);
    return 0;
}

// LDS bank conflicts of the warp kernel's byte taps by window pitch and lane -> row mapping (gfx950).
// A wavefront = a 32 x 8 footprint, lane l holds 4 pixels (12 bytes) of one row; 12 byte loads per pixel as in gather_blend_sums.
//   pitch 160 (wide window, 10 chunks per row) / 112 (COMPACT window, 7 chunks per row)
//   rows natural   : row = l >> 3                       (lanes 0-31 = rows 0-3)
//   rows interleave: row = 2 * ((l >> 3) & 3) + (l >> 5)  (lanes 0-31 = rows 0, 2, 4, 6)
// `shear` adds row * shear bytes to a row's start (a rotated footprint); `sub` is the byte offset inside the dword.
// Prints cycles per wave64 LDS instruction per CU (all four SIMDs issuing, 8 wavefronts per SIMD).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int PITCH, bool INTERLEAVE>
__global__ __launch_bounds__(64) void k(uint32_t* out, int reps, int shear, int sub)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t l = threadIdx.x;
    const uint32_t row = INTERLEAVE ? 2u * ((l >> 3) & 3u) + (l >> 5) : (l >> 3);
    const uint32_t base = (uint32_t)(uintptr_t)&s[0] + (l & 7) * 12u + row * (uint32_t)PITCH + row * (uint32_t)shear + (uint32_t)sub;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0, a9 = 0, a10 = 0, a11 = 0;
    for (int r = 0; r < reps; ++r) {
        const uint32_t at = base + (uint32_t)(r & 3) * 3u;
        if (PITCH == 160)
            asm volatile("ds_read_u8 %0, %12 offset:0\n\tds_read_u8_d16_hi %1, %12 offset:3\n\tds_read_u8 %2, %12 offset:1\n\tds_read_u8_d16_hi %3, %12 offset:4\n\t"
                         "ds_read_u8 %4, %12 offset:2\n\tds_read_u8_d16_hi %5, %12 offset:5\n\tds_read_u8 %6, %12 offset:160\n\tds_read_u8_d16_hi %7, %12 offset:163\n\t"
                         "ds_read_u8 %8, %12 offset:161\n\tds_read_u8_d16_hi %9, %12 offset:164\n\tds_read_u8 %10, %12 offset:162\n\tds_read_u8_d16_hi %11, %12 offset:165\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7), "=&v"(a8), "=&v"(a9), "=&v"(a10), "=&v"(a11) : "v"(at));
        else
            asm volatile("ds_read_u8 %0, %12 offset:0\n\tds_read_u8_d16_hi %1, %12 offset:3\n\tds_read_u8 %2, %12 offset:1\n\tds_read_u8_d16_hi %3, %12 offset:4\n\t"
                         "ds_read_u8 %4, %12 offset:2\n\tds_read_u8_d16_hi %5, %12 offset:5\n\tds_read_u8 %6, %12 offset:112\n\tds_read_u8_d16_hi %7, %12 offset:115\n\t"
                         "ds_read_u8 %8, %12 offset:113\n\tds_read_u8_d16_hi %9, %12 offset:116\n\tds_read_u8 %10, %12 offset:114\n\tds_read_u8_d16_hi %11, %12 offset:117\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7), "=&v"(a8), "=&v"(a9), "=&v"(a10), "=&v"(a11) : "v"(at));
        asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(a8), "v"(a9), "v"(a10), "v"(a11));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11;
}

template <int PITCH, bool INTERLEAVE> void run(const char* name, int shear, int sub)
{
    uint32_t* out;
    const int blocks = 256 * 32 * 8;
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    k<PITCH, INTERLEAVE><<<blocks, 64>>>(out, 10, shear, sub);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 3000;
    hipEventRecord(e0); k<PITCH, INTERLEAVE><<<blocks, 64>>>(out, reps, shear, sub); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)blocks / 256 * reps * 12;
    printf("%-40s shear %2d sub %d: %.3f ms -> %.2f cycles per wave64 LDS instruction per CU @2.1 GHz\n", name, shear, sub, ms, ms * 1e-3 * 2.1e9 / instr_per_cu);
    hipFree(out);
}
int main()
{
    const int shears[] = { 0, 1, 3, -1 & 0xFF };
    for (int sub = 0; sub < 4; sub += 1)
        for (int si = 0; si < 3; ++si) {
            const int sh = shears[si];
            run<160, false>("pitch 160 rows natural", sh, sub);
            run<160, true>("pitch 160 rows interleaved", sh, sub);
            run<112, false>("pitch 112 rows natural", sh, sub);
            run<112, true>("pitch 112 rows interleaved", sh, sub);
        }
    return 0;
}

// LDS throughput of the access shapes of the warp kernel's tap fetch on gfx950: cycles per wave64 instruction per CU with all four
// SIMDs of every CU issuing (8 wavefronts per SIMD), lane stride 12 bytes (4 pixels of 3 bytes), rows of 160 bytes every 8 lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)&s[0] + (threadIdx.x & 7) * 12u + (threadIdx.x >> 3) * 160u;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    for (int r = 0; r < reps; ++r) {
        const uint32_t at = base + (uint32_t)(r & 7) * 3u;
        if (KIND == 0)        // 8 byte loads (u8 / u8_d16_hi alternating)
            asm volatile("ds_read_u8 %0, %8 offset:0\n\tds_read_u8_d16_hi %1, %8 offset:3\n\tds_read_u8 %2, %8 offset:1\n\tds_read_u8_d16_hi %3, %8 offset:4\n\t"
                         "ds_read_u8 %4, %8 offset:2\n\tds_read_u8_d16_hi %5, %8 offset:5\n\tds_read_u8 %6, %8 offset:160\n\tds_read_u8_d16_hi %7, %8 offset:163\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(at));
        else if (KIND == 1)   // 8 aligned dword loads
            asm volatile("ds_read_b32 %0, %8 offset:0\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:160\n\t"
                         "ds_read_b32 %4, %8 offset:164\n\tds_read_b32 %5, %8 offset:168\n\tds_read_b32 %6, %8 offset:320\n\tds_read_b32 %7, %8 offset:324\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(at & ~3u));
        else if (KIND == 2)   // 8 u16 loads
            asm volatile("ds_read_u16 %0, %8 offset:0\n\tds_read_u16 %1, %8 offset:2\n\tds_read_u16 %2, %8 offset:4\n\tds_read_u16 %3, %8 offset:6\n\t"
                         "ds_read_u16 %4, %8 offset:160\n\tds_read_u16 %5, %8 offset:162\n\tds_read_u16 %6, %8 offset:164\n\tds_read_u16 %7, %8 offset:166\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(at & ~1u));
        else if (KIND == 3)   // 8 unaligned dword loads (byte-granular address)
            asm volatile("ds_read_b32 %0, %8 offset:0\n\tds_read_b32 %1, %8 offset:3\n\tds_read_b32 %2, %8 offset:6\n\tds_read_b32 %3, %8 offset:160\n\t"
                         "ds_read_b32 %4, %8 offset:163\n\tds_read_b32 %5, %8 offset:166\n\tds_read_b32 %6, %8 offset:320\n\tds_read_b32 %7, %8 offset:323\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(at));
        asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND> void run(const char* name)
{
    uint32_t* out;
    const int blocks = 256 * 32 * 8;               // 8 generations of 32 single-wave workgroups per CU
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    k<KIND><<<blocks, 64>>>(out, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 4000;
    hipEventRecord(e0); k<KIND><<<blocks, 64>>>(out, reps); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)blocks / 256 * reps * 8;
    printf("%-44s %.3f ms -> %.2f cycles per wave64 instruction per CU @2.1 GHz\n", name, ms, ms * 1e-3 * 2.1e9 / instr_per_cu);
    hipFree(out);
}
int main()
{
    run<0>("ds_read_u8 / ds_read_u8_d16_hi (byte taps)");
    run<1>("ds_read_b32 aligned");
    run<2>("ds_read_u16 (2-byte aligned)");
    run<3>("ds_read_b32 byte-unaligned");
    return 0;
}

// Can the warp kernel's taps be fetched as 16-bit LDS loads (6 per pixel instead of 12 byte loads)?  Cycles per wave64 instruction and CU of
// ds_read_u16 at byte offsets 0..3 inside a dword (a tap row starts at byte 3 ix: every alignment occurs), lane stride 12 bytes, rows of
// 160 bytes every 8 lanes -- against the byte loads of today.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, int reps, int sub)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)&s[0] + (threadIdx.x & 7) * 12u + (threadIdx.x >> 3) * 160u + (uint32_t)sub;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
    for (int r = 0; r < reps; ++r) {
        const uint32_t at = base + (uint32_t)(r & 3) * 12u * 0u;
        if (KIND == 0)        // six u16 loads: rows iy and iy + 1, bytes 0-1, 2-3, 4-5
            asm volatile("ds_read_u16 %0, %6 offset:0\n\tds_read_u16 %1, %6 offset:2\n\tds_read_u16 %2, %6 offset:4\n\t"
                         "ds_read_u16 %3, %6 offset:160\n\tds_read_u16 %4, %6 offset:162\n\tds_read_u16 %5, %6 offset:164\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(at));
        else                  // six byte loads (half a pixel's taps of today)
            asm volatile("ds_read_u8 %0, %6 offset:0\n\tds_read_u8_d16_hi %1, %6 offset:3\n\tds_read_u8 %2, %6 offset:1\n\t"
                         "ds_read_u8_d16_hi %3, %6 offset:4\n\tds_read_u8 %4, %6 offset:2\n\tds_read_u8_d16_hi %5, %6 offset:5\n\t"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(at));
        asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5;
}

template <int KIND> void run(const char* name, int sub)
{
    uint32_t* out;
    const int blocks = 256 * 32 * 8;
    hipMalloc(&out, (size_t)blocks * 64 * 4);
    k<KIND><<<blocks, 64>>>(out, 10, sub);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 3000;
    hipEventRecord(e0); k<KIND><<<blocks, 64>>>(out, reps, sub); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)blocks / 256 * reps * 6;
    printf("%-36s byte offset %d in the dword: %.3f ms -> %.2f cycles per wave64 LDS instruction per CU @2.1 GHz\n", name, sub, ms, ms * 1e-3 * 2.1e9 / instr_per_cu);
    hipFree(out);
}
int main()
{
    for (int sub = 0; sub < 4; ++sub) { run<0>("ds_read_u16 x 6", sub); run<1>("ds_read_u8 / _d16_hi x 6", sub); }
    return 0;
}

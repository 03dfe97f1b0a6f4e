// Are byte-unaligned 8-byte LDS reads correct and how fast are they on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int STRIDE, int OFF>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int* bad, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) s[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    unsigned long long acc = 0;
    int nbad = 0;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = (threadIdx.x * STRIDE + OFF + u * 1536 + (r & 15) * 8) & 16383 & ~(STRIDE == 8 ? 7 : 0);
            const int oo = o > 16376 ? 16376 - 8 : o;
            unsigned long long v;
            __builtin_memcpy(&v, s + oo, 8);
            acc += v;
            if (r == 0) {
                unsigned long long e = 0;
                for (int b = 7; b >= 0; --b) e = (e << 8) | (uint8_t)((oo + b) * 7 + 3);
                nbad += e != v;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (nbad) atomicAdd(bad, nbad);
}
template <int STRIDE, int OFF> void run(const char* name)
{
    unsigned long long* out; int* bad;
    hipMalloc(&out, 2048 * 256 * 8); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    k<STRIDE, OFF><<<2048, 256>>>(out, bad, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<STRIDE, OFF><<<2048, 256>>>(out, bad, 2000); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    const double reads = 2048.0 * 4 * 2000 * 8;      // wave-instructions
    printf("%-34s mismatches=%d  %.3f ms  -> %.2f cycles per wave ds_read_b64 per CU @2.1GHz\n", name, hb, ms,
           ms * 1e-3 * 2.1e9 * 256 / reads);
}
int main()
{
    run<8, 0>("aligned 8B, stride 8");
    run<3, 0>("unaligned 8B, stride 3 (pixels)");
    run<3, 1>("unaligned 8B, stride 3, +1");
    run<12, 2>("unaligned 8B, stride 12, +2");
    return 0;
}

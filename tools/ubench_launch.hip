// Wave-launch throughput of the MI355X for one-wavefront workgroups with the warp kernel's resource footprint.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/ubench_launch && tools/ubench_launch
// Each workgroup: 64 threads, 64 VGPRs (forced by clobbering v63), `LDS` bytes of LDS; body = `SLEEP` x s_sleep 127 (~8k cycles
// each... see table) or nothing, then one dword store by lane 0 (so that the launch is not optimised away).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS, int SLEEP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned* out)
{
    __shared__ unsigned s[LDS / 4];
    if (threadIdx.x == 0) s[0] = blockIdx.x;
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
#pragma unroll
    for (int i = 0; i < SLEEP; ++i) asm volatile("s_sleep 64");        // 64 x 64 cycles
    if (threadIdx.x == 0) out[blockIdx.x & 1023] = s[0];
}

template <int LDS, int SLEEP, int WAVES>
void run(const char* name, unsigned* d, int nwaves)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = nwaves / WAVES;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<LDS, SLEEP, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, d);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<LDS, SLEEP, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-44s %8d waves  %.4f ms  %.2f waves/ns  %.1f ns per wave and XCD\n", name, nwaves, ms, nwaves / (ms * 1e6), ms * 1e6 * 8 / nwaves);
}

// Shader clock under load: every wave runs `iters` rounds of the warp kernel's instruction mix (f64 fma chain + 24-bit integer
// mads) and lane 0 reads s_memtime (shader clock) and s_memrealtime (100 MHz) around it.
__global__ __launch_bounds__(256) void clock_kernel(unsigned long long* out, int iters, double seed)
{
    double a = seed + threadIdx.x, b = 1.000001, c = 0.5;
    unsigned x = threadIdx.x, y = 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a = __builtin_fma(a, b, c);
            c = __builtin_fma(c, b, a);
            x = __umul24(x, y) + 7u;
            y = __umul24(y, x) + 3u;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (a == 1.25 && x == 77u) out[3] = y + (unsigned long long)c;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int main()
{
    {
        unsigned long long* d; hipMalloc(&d, 64);
        for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(clock_kernel, dim3(256 * 8), dim3(256), 0, 0, d, 20000, 0.1);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("shader clock under an f64-fma + int-mad load on every SIMD (8 waves): %llu cycles in %llu ticks of 100 MHz = %.3f GHz\n",
               h[0], h[1], (double)h[0] / (double)h[1] * 0.1);
    }
    unsigned* d; hipMalloc(&d, 4096);
    const int n = 2430000 / 4 * 4;
    run<2832, 0, 1>("1 wave/WG, 2.8 KB LDS, empty", d, n);
    run<4944, 0, 1>("1 wave/WG, 4.9 KB LDS, empty", d, n);
    run<64, 0, 1>("1 wave/WG, 64 B LDS, empty", d, n);
    run<2832 * 4, 0, 4>("4 waves/WG, 11 KB LDS, empty", d, n);
    run<2832, 1, 1>("1 wave/WG, 2.8 KB LDS, sleep 4k cycles", d, n);
    run<2832, 2, 1>("1 wave/WG, 2.8 KB LDS, sleep 8k cycles", d, n);
    run<2832 * 4, 2, 4>("4 waves/WG, 11 KB LDS, sleep 8k cycles", d, n);
    run<2832, 4, 1>("1 wave/WG, 2.8 KB LDS, sleep 16k cycles", d, n);
    return 0;
}

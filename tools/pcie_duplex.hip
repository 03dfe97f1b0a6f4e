// PCIe duplex probe: how fast can 1 GiB go up while 1 GiB comes down, by which engines?
//   sdma/sdma    hipMemcpyAsync both ways (pinned host memory) -- what csrc/hostpipe.hip does today
//   kern/sdma    host -> device by a copy KERNEL reading mapped pinned host memory, device -> host by hipMemcpyAsync
//   sdma/kern    the other way round (kernel stores to mapped host memory)
//   kern/kern    both by kernels (two streams)
// Each also alone.  GB/s per direction, best of 4.  Build: make -C tools pcie_duplex; run: tools/pcie_duplex [MiB] [blocks]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 20;
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    const size_t piece = 64u << 20;
    uint8_t *h_up, *h_dn, *d_up, *d_dn;
    CK(hipHostMalloc(&h_up, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_dn, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_up, bytes));
    CK(hipMalloc(&d_dn, bytes));
    memset(h_up, 1, bytes);
    memset(h_dn, 0, bytes);
    CK(hipMemset(d_dn, 2, bytes));
    hipStream_t su[4], sd[4];
    for (auto& s : su) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& s : sd) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const char* names[2] = { "sdma", "kern" };
    printf("%zu MiB each way, copy kernel: %d blocks of 256 threads per 64 MiB piece\n", bytes >> 20, blocks);
    for (int um = -1; um < 2; ++um)
        for (int dm = -1; dm < 2; ++dm) {
            if (um < 0 && dm < 0) continue;
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                int k = 0;
                for (size_t off = 0; off < bytes; off += piece, ++k) {
                    const size_t m = bytes - off < piece ? bytes - off : piece;
                    if (um == 0) CK(hipMemcpyAsync(d_up + off, h_up + off, m, hipMemcpyHostToDevice, su[k & 3]));
                    if (um == 1) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, su[k & 3], (const u4*)(h_up + off), (u4*)(d_up + off), m / 16);
                    if (dm == 0) CK(hipMemcpyAsync(h_dn + off, d_dn + off, m, hipMemcpyDeviceToHost, sd[k & 3]));
                    if (dm == 1) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, sd[k & 3], (const u4*)(d_dn + off), (u4*)(h_dn + off), m / 16);
                }
                CK(hipDeviceSynchronize());
                const double dt = now() - t0;
                if (rep > 0 && dt < best) best = dt;
            }
            printf("up %-5s down %-5s : %6.1f GB/s per direction (%.1f ms)\n", um < 0 ? "-" : names[um], dm < 0 ? "-" : names[dm], bytes / best / 1e9, best * 1e3);
        }
    // sanity: what came down / went up
    CK(hipMemcpy(h_dn, d_up, 64, hipMemcpyDeviceToHost));
    printf("check: up byte %d (1), down byte %d (2)\n", h_dn[0], h_dn[bytes - 1] == 2 ? 2 : h_dn[bytes - 1]);
    return 0;
}

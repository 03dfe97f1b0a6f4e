// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the warp kernel.
// Every kernel touches each of the N source bytes exactly once from HBM (N = 1,866,240,000: one config-2
// clip, far beyond the 256 MiB Infinity Cache) and the first two write N bytes:
//   calib_wide16   : 16-byte loads and stores per lane (plain coalesced copy)
//   calib_gather8  : two unaligned 8-byte loads per "pixel" at byte offset 3*p (the warp kernel's tap
//                    fetch: rows iy and iy+1), 12-byte store per 4 pixels (the warp kernel's store)
//   calib_read8    : the same loads, no stores
//   calib_stage16  : the warp kernel's CURRENT tap fetch: per wavefront two dword-aligned 16-byte global->LDS loads per
//                    lane (global_load_lds_dwordx4) covering 12.8 rows x 160 bytes around its 32 x 8 footprint (windows
//                    of neighbouring footprints overlap, as in the kernel), 12-byte store per 4 pixels
// Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   and   --pmc WRITE_SIZE ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void calib_wide16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <bool STORE>
__global__ __launch_bounds__(256) void calib_gather8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H, int nfr)
{
    // same tiling as the warp kernel: 128 x 16 tiles, lane = 4 consecutive pixels, identity coordinates
    const int f = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * 128 + wave * 32 + (lane & 7) * 4;
    const uint8_t* src = in + (size_t)f * W * H * 3;
    uint8_t* dst = out + (size_t)f * W * H * 3;
    for (int q = 0; q < 2; ++q) {
        const int y = blockIdx.y * 16 + q * 8 + (lane >> 3);
        if (y >= H - 1 || x0 + 6 >= W) continue;
        uint32_t acc[4];
        for (int j = 0; j < 4; ++j) {
            const uint32_t o = ((uint32_t)y * W + x0 + j) * 3u;
            uint2 a, b;
            __builtin_memcpy(&a, src + o, 8);
            __builtin_memcpy(&b, src + o + 3u * W, 8);
            acc[j] = (a.x ^ a.y ^ b.x ^ b.y) & 0xFFFFFFu;
        }
        if (STORE) {
            uint3 d; d.x = acc[0] | (acc[1] << 24); d.y = (acc[1] >> 8) | (acc[2] << 16); d.z = (acc[2] >> 16) | (acc[3] << 8);
            *reinterpret_cast<uint3*>(dst + ((size_t)y * W + x0) * 3) = d;
        } else if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) dst[0] = 1;
    }
}

__global__ __launch_bounds__(256) void calib_stage16(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H, int nfr)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[4][2048 + 64];
    const int f = blockIdx.z, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int xa = blockIdx.x * 128 + wave * 32, ya = blockIdx.y * 8;
    const int x0 = xa + (lane & 7) * 4, y = ya + (lane >> 3);
    const uint8_t* src = in + (size_t)f * W * H * 3;
    uint8_t* dst = out + (size_t)f * W * H * 3;
    const uint32_t row_bytes = 3u * W;
    const int sx0 = min(max(xa - 2, 0), (int)(row_bytes - 160) / 3), sy0 = min(max(ya - 2, 0), H - 13);   // identity warp + slack
    const uint32_t bs = (3u * sx0) & ~3u;
    const uint8_t* gbase = src + (size_t)sy0 * row_bytes + bs;
    uint32_t o0 = __umul24(((uint32_t)lane * 205u) >> 11, row_bytes - 160u) + ((uint32_t)lane << 4);
    uint32_t o1 = __umul24((((uint32_t)lane + 64u) * 205u) >> 11, row_bytes - 160u) + (((uint32_t)lane << 4) + 1024u);
    asm("" : "+v"(o0));
    asm("" : "+v"(o1));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o0),
                                     (__attribute__((address_space(3))) void*)&s_src[wave][0], 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o1),
                                     (__attribute__((address_space(3))) void*)&s_src[wave][1024], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (y >= H || x0 + 3 >= W) return;
    uint32_t acc[4];
    for (int j = 0; j < 4; ++j) {
        const uint32_t at = (uint32_t)(y - sy0) * 160u + 3u * (uint32_t)(x0 + j) - bs;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(&s_src[wave][at & ~3u]);
        acc[j] = (__builtin_amdgcn_alignbyte(p[1], p[0], at) ^ __builtin_amdgcn_alignbyte(p[41], p[40], at)) & 0xFFFFFFu;
    }
    uint3 d; d.x = acc[0] | (acc[1] << 24); d.y = (acc[1] >> 8) | (acc[2] << 16); d.z = (acc[2] >> 16) | (acc[3] << 8);
    *reinterpret_cast<uint3*>(dst + ((size_t)y * W + x0) * 3) = d;
}

int main()
{
    const int W = 1920, H = 1080, n = 300;
    const size_t N = (size_t)W * H * 3 * n;
    uint8_t *in, *out;
    hipMalloc(&in, N + 64); hipMalloc(&out, N + 64);
    hipMemset(in, 0x5A, N + 64); hipMemset(out, 0, N + 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        calib_wide16<<<4096, 256>>>((const uint4*)in, (uint4*)out, N / 16);
        calib_gather8<true><<<dim3(15, 68, n), 256>>>(in, out, W, H, n);
        calib_gather8<false><<<dim3(15, 68, n), 256>>>(in, out, W, H, n);
        calib_stage16<<<dim3(15, 135, n), 256>>>(in, out, W, H, n);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: read %zu, written %zu (wide16, gather8<store>, stage16), 0 (gather8<read only>)\n", N, N);
    return 0;
}

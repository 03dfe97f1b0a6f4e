// What two gfx950 instructions DO (not how fast they are), for the round-6 trims of the warp kernel's blend and coordinate chain:
//   1. v_pk_mad_u16 with the clamp bit: does it SATURATE each 16-bit half at 65535?  (the blend's weight 64 (32 - fx) (32 - fy) is 65536
//      for fx = fy = 0 and has to become 65535 there.)  Also op_sel_hi = [1, 0, 1]: the low half of src1 used for both halves.
//   3. v_mfma_f64_16x16x4_f64: its issue cost (is the float64 matrix rate of gfx950 above the vector rate?  No: 64 cycles for 1,024 multiply-adds).
//   2. v_rcp_f64: its largest relative error over 2^26 evenly spaced mantissas (times a few exponents), and the error left after ONE
//      Newton step r' = r + r (1 - x r) -- what the cheap coordinate chain would keep if it dropped its second step.
// Build: make -C tools ubench_semantics      Run: tools/ubench_semantics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>

__global__ void pk_clamp_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r0, r1, r2;
    const uint32_t zero = 0;
    asm volatile("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(r0) : "v"(a[i]), "v"(b[i]), "v"(zero));
    asm volatile("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1] clamp" : "=v"(r1) : "v"(a[i]), "v"(b[i]), "v"(zero));
    asm volatile("v_pk_mul_lo_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r2) : "v"(a[i]), "v"(b[i]));
    out[3 * i] = r0;
    out[3 * i + 1] = r1;
    out[3 * i + 2] = r2;
}

// max over the grid of |1 - x r| for r = v_rcp_f64(x) and for r after one Newton step, as raw double bits (positive doubles order as integers)
__global__ void rcp_error_kernel(unsigned long long* __restrict__ worst, int exp_lo, int exp_hi, unsigned shift)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t mant = (idx << shift) | ((idx * 0x9E3779B97F4A7C15ull) >> (64 - shift));     // evenly spaced top bits, hashed low bits
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    for (int ex = exp_lo; ex <= exp_hi; ++ex) {
        const double x = __longlong_as_double(((uint64_t)(1023 + ex) << 52) | (mant & 0xFFFFFFFFFFFFFull));
        double r = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, r, 1.0);
        e0 = fmax(e0, fabs(e));
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-x, r, 1.0);
        e1 = fmax(e1, fabs(e));
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-x, r, 1.0);
        e2 = fmax(e2, fabs(e));
    }
    atomicMax(&worst[0], (unsigned long long)__double_as_longlong(e0));
    atomicMax(&worst[1], (unsigned long long)__double_as_longlong(e1));
    atomicMax(&worst[2], (unsigned long long)__double_as_longlong(e2));
}

// issue cost of v_mfma_f64_16x16x4_f64 (1,024 multiply-adds): 4 independent accumulators, 256 instructions per wavefront, one wavefront per SIMD
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void mfma_f64_kernel(double* out, unsigned long long* ticks)
{
    d4 acc[4] = {};
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    {
        // a = 64 (32 - fx) | 64 fx << 16, b = (32 - fy) | junk << 16: every (fx, fy)
        const int n = 32 * 32;
        uint32_t ha[n], hb[n], hout[3 * n];
        for (int fx = 0; fx < 32; ++fx)
            for (int fy = 0; fy < 32; ++fy) {
                ha[fx * 32 + fy] = (64u * (32u - fx)) | ((64u * fx) << 16);
                hb[fx * 32 + fy] = (32u - fy) | (0xABCDu << 16);
            }
        uint32_t *da, *db, *dout;
        CHECK(hipMalloc(&da, sizeof ha)); CHECK(hipMalloc(&db, sizeof hb)); CHECK(hipMalloc(&dout, sizeof hout));
        CHECK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
        pk_clamp_kernel<<<(n + 63) / 64, 64>>>(da, db, dout, n);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost));
        int bad_plain = 0, bad_sel = 0, bad_mul = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t alo = ha[i] & 0xFFFF, ahi = ha[i] >> 16, blo = hb[i] & 0xFFFF, bhi = hb[i] >> 16;
            const uint32_t sat_lo = alo * blo > 65535 ? 65535 : alo * blo;
            const uint32_t plain = sat_lo | ((ahi * bhi > 65535 ? 65535 : ahi * bhi) << 16);
            const uint32_t sel = sat_lo | ((ahi * blo > 65535 ? 65535 : ahi * blo) << 16);
            const uint32_t mul = ((alo * blo) & 0xFFFF) | (((ahi * blo) & 0xFFFF) << 16);
            bad_plain += hout[3 * i] != plain;
            bad_sel += hout[3 * i + 1] != sel;
            bad_mul += hout[3 * i + 2] != mul;
            if (i == 0) printf("fx = fy = 0: a = %08x b = %08x -> pk_mad clamp %08x, with op_sel_hi [1,0,1] %08x, pk_mul_lo op_sel_hi [1,0] %08x\n", ha[i], hb[i], hout[0], hout[1], hout[2]);
        }
        printf("v_pk_mad_u16 clamp: %d of %d differ from per-half saturation; with op_sel_hi:[1,0,1] (low half of src1 for both halves): %d differ; v_pk_mul_lo_u16 op_sel_hi:[1,0]: %d differ from the wrapping product\n",
               bad_plain, n, bad_sel, bad_mul);
    }
    {
        unsigned long long* dw;
        CHECK(hipMalloc(&dw, 3 * sizeof(unsigned long long)));
        CHECK(hipMemset(dw, 0, 3 * sizeof(unsigned long long)));
        const unsigned bits = 26;
        rcp_error_kernel<<<(1u << bits) / 256, 256>>>(dw, -4, 3, 52 - bits);
        CHECK(hipDeviceSynchronize());
        unsigned long long hw[3];
        CHECK(hipMemcpy(hw, dw, sizeof hw, hipMemcpyDeviceToHost));
        double e[3];
        for (int k = 0; k < 3; ++k) { memcpy(&e[k], &hw[k], 8); }
        printf("v_rcp_f64 over 2^%u mantissas x exponents 2^-4..2^3: max |1 - x r| = %.4g = 2^%.2f; after one Newton step %.4g = 2^%.2f (%.1f ulp of 2^-52); after two %.4g = 2^%.2f\n",
               bits, e[0], log2(e[0]), e[1], log2(e[1]), e[1] / ldexp(1.0, -52), e[2], log2(e[2]));
    }
    {
        double* dout; unsigned long long* dt;
        const int blocks = 1024;                                     // one wavefront per SIMD
        CHECK(hipMalloc(&dout, blocks * 64 * sizeof(double))); CHECK(hipMalloc(&dt, blocks * sizeof(unsigned long long)));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        mfma_f64_kernel<<<blocks, 64>>>(dout, dt);
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; ++r) mfma_f64_kernel<<<blocks, 64>>>(dout, dt);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        static unsigned long long ht[1024];
        CHECK(hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost));
        double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)ht[i]; mean /= blocks;
        const double flops = 20.0 * blocks * 256.0 * 2.0 * 16 * 16 * 4;
        printf("v_mfma_f64_16x16x4_f64: %.1f shader cycles per instruction (256 back to back per wavefront, 4 accumulators) = %.1f multiply-adds per cycle and SIMD (the vector unit: 16); "
               "20 short launches, one wavefront per SIMD: %.1f TFLOP/s float64 (vector float64 peak: 78.6)\n",
               mean / 256.0, 1024.0 / (mean / 256.0), flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}

// Micro-benchmark: issue cost (cycles per wave64 instruction) of the VALU ops the warp kernel uses.
// One wave per SIMD-ish; each op repeated in an unrolled chain of 8 independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 20000

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, const double* in, long long* cyc)
{
    double a[8]; float fa[8]; uint32_t ia[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[i + threadIdx.x % 3]; fa[i] = (float)a[i]; ia[i] = (uint32_t)(a[i] * 1000) + threadIdx.x; }
    double b[8];
    for (int i = 0; i < 8; ++i) b[i] = in[(i + threadIdx.x) % 7] * 1e-3;
    double c = in[9], d = in[10]; float fc = (float)c; uint32_t ic = (uint32_t)(c * 77), id = (uint32_t)(d * 55);
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], c, d);
            if (OP == 1) a[i] = a[i] * c;
            if (OP == 2) a[i] = a[i] + c;
            if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 4) { fa[i] = (float)a[i]; a[i] = a[i] + 1.0; }          // cvt_f32_f64 + add_f64
            if (OP == 5) ia[i] = __umul24(ia[i], ic);
            if (OP == 6) ia[i] = __umul24(ia[i], ic) + id;                      // mad_u32_u24
            if (OP == 7) ia[i] = ia[i] * ic;                                    // mul_lo_u32
            if (OP == 8) ia[i] = (ia[i] & ic) | id;                             // and_or
            if (OP == 9) ia[i] = ia[i] > ic ? id : ia[i] + 1;                   // cmp + cndmask (+add)
            if (OP == 10) fa[i] = fa[i] * fc;                                   // mul_f32
            if (OP == 11) fa[i] = rintf(fa[i]) + fc;                            // rndne + add f32
            if (OP == 12) ia[i] = __builtin_amdgcn_alignbit(ia[i], ic, 24);
            if (OP == 13) ia[i] = (ia[i] >> 8) & 255u;                          // bfe
            if (OP == 14) a[i] = fmin(a[i], c);
            if (OP == 15) ia[i] = (uint32_t)((int)rintf(fa[i])) + ia[i];         // rndne+cvt_i32_f32+add
            if (OP == 16) a[i] = (double)ia[i] + a[i];                          // cvt_f64_u32 + add_f64
            if (OP == 17) ia[i] = __builtin_amdgcn_perm(ia[i], ic, 0x02010007);
            if (OP == 18) a[i] = __builtin_fma(c, b[i], a[i]);                  // v_fmac_f64 acc, sgpr, vgpr (the Jacobi inner op)
            if (OP == 19) a[i] = __builtin_fma(b[(i + 1) & 7], b[i], a[i]);     // v_fmac_f64 acc, vgpr, vgpr
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + fa[i] + ia[i] + b[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char* name, int nops, int waves_per_simd)
{
    double *out, *in; long long* cyc;
    int blocks = 1024 * waves_per_simd;   // 256 CUs x 4 SIMDs
    hipMalloc(&out, blocks * 64 * 8); hipMalloc(&in, 16 * 8); hipMalloc(&cyc, blocks * 8);
    double h[16]; for (int i = 0; i < 16; ++i) h[i] = 1.0 + i * 0.001;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    k<OP><<<blocks, 64>>>(out, in, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<OP><<<blocks, 64>>>(out, in, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long hc[8]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    double per = (double)hc[0] / (REP * 8.0 * nops);
    // memtime ticks at 100 MHz? report both raw ticks/instr and wall-based cycles assuming all SIMDs busy
    double wall_cyc_per_instr = ms * 1e-3 * 2.4e9 / (REP * 8.0 * nops * waves_per_simd);
    printf("%-28s waves/SIMD=%d  memtime ticks/instr=%.3f  wall: %.3f ms -> %.2f cyc/instr @2.4GHz\n", name, waves_per_simd, per, ms, wall_cyc_per_instr);
    hipFree(out); hipFree(in); hipFree(cyc);
}

int main()
{
    for (int w = 2; w <= 8; w *= 4) {
        run<0>("v_fma_f64", 1, w); run<1>("v_mul_f64", 1, w); run<2>("v_add_f64", 1, w); run<3>("v_rcp_f64", 1, w);
        run<4>("cvt_f32_f64+add_f64", 2, w); run<5>("v_mul_u32_u24", 1, w); run<6>("v_mad_u32_u24", 1, w);
        run<7>("v_mul_lo_u32", 1, w); run<8>("v_and_or_b32", 1, w); run<9>("cmp+cndmask+add", 3, w);
        run<10>("v_mul_f32", 1, w); run<11>("rndne_f32+add_f32", 2, w); run<12>("v_alignbit", 1, w);
        run<13>("v_bfe_u32", 1, w); run<14>("v_min_f64", 1, w); run<15>("rndne+cvt_i32_f32+add", 3, w);
        run<16>("cvt_f64_u32+add_f64", 2, w); run<17>("v_perm_b32", 1, w);
        run<18>("v_fmac_f64 acc,sgpr,vgpr", 1, w); run<19>("v_fmac_f64 acc,vgpr,vgpr", 1, w);
    }
    return 0;
}

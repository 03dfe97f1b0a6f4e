// Kernel 1: Jacobi temporal smoothing of every vertex path (reference: meshflowstabilizer.py:844-878,
// looped over vertices at :695-704).  Each of the S = V*2 series is an independent 1-D problem
//   x_new[t] = inv_on[t] * (b[t] + 2*lam[t] * sum_{d=-W..W} taps[d] * x[t+d])        (zero halo)
// iterated `iters` times from x = b.  float64 throughout: after N sweeps the reference is far from
// converged, so the iteration itself is replicated, and an fp32 state misses the 1e-4 bar.
//
// Mapping (gfx950): one workgroup of WAVES wavefronts per series.  Thread l owns K consecutive frames
// [l*K, l*K+K); the state x lives in LDS (double-buffered, OMEGA zero-halo entries on both sides);
// each sweep a thread pulls its K+2*OMEGA window into registers with ds_read_b64, forms K dot products
// of 2*OMEGA+1 taps with v_fma_f64 (taps are wave-uniform -> scalar registers), and writes K values
// back; one barrier per sweep.  Everything stays on chip for all sweeps: HBM traffic is one read of b and
// one write of x.  This kernel is FP64-VALU / LDS bound, not HBM bound.  K is kept minimal (5 frames per thread at
// OMEGA = 10, 10 at OMEGA = 30); clips longer than 64 K frames spread each series over 2-8 wavefronts.
#include <stdlib.h>

#include "mf_common.h"

namespace mf {

template <int OMEGA, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void jacobi_wave_kernel(const double* __restrict__ b, double* __restrict__ x_out,
                                                         const double* __restrict__ taps,
                                                         const double* __restrict__ lam,
                                                         const double* __restrict__ inv_on, int F, int S, int iters)
{
    constexpr int NT = 2 * OMEGA + 1;
    constexpr int NTHR = 64 * WAVES;
    constexpr int LEN = NTHR * K + 2 * OMEGA;
    __shared__ double xs[2][LEN];
    const int s = blockIdx.x;
    const int lane = threadIdx.x;

    double w[NT];
#pragma unroll
    for (int d = 0; d < NT; ++d) w[d] = taps[d];

    double bt[K], two_lam[K], inv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = lane * K + k;
        const bool in = t < F;
        bt[k] = in ? b[(size_t)t * S + s] : 0.0;
        two_lam[k] = in ? 2.0 * lam[t] : 0.0;
        inv[k] = in ? inv_on[t] : 0.0;            // frames past the end stay exactly 0 = the zero halo
    }
    for (int i = lane; i < LEN; i += NTHR) { xs[0][i] = 0.0; xs[1][i] = 0.0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) xs[0][OMEGA + lane * K + k] = bt[k];   // x_start = b
    __syncthreads();

    int cur = 0;
    double xn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) xn[k] = bt[k];
    for (int it = 0; it < iters; ++it) {
        double win[K + 2 * OMEGA];
        const double* src = &xs[cur][lane * K];
#pragma unroll
        for (int j = 0; j < K + 2 * OMEGA; ++j) win[j] = src[j];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < NT; ++d) acc = __builtin_fma(w[d], win[k + d], acc);
            xn[k] = inv[k] * __builtin_fma(two_lam[k], acc, bt[k]);
        }
        double* dst = &xs[cur ^ 1][OMEGA + lane * K];
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = xn[k];
        cur ^= 1;
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = lane * K + k;
        if (t < F) x_out[(size_t)t * S + s] = xn[k];
    }
}

// Fallback for any other (omega, F) -- specialised radii: 5, 10, 15, 20, 30 -- one 256-thread workgroup per series,
// thread-per-frame loop over LDS.  Same summation order as the specialised kernel, so both give identical bits, at a tenth of
// the speed (F = 300, 16 x 16 mesh, 100 sweeps: omega = 40: 693 us = 4.1 TFLOP/s against 48 us = 16 TFLOP/s for omega = 10).
__global__ __launch_bounds__(256) void jacobi_generic_kernel(const double* __restrict__ b, double* __restrict__ x_out,
                                                             const double* __restrict__ taps,
                                                             const double* __restrict__ lam,
                                                             const double* __restrict__ inv_on, int F, int S,
                                                             int omega, int iters)
{
    extern __shared__ double dyn[];
    const int len = F + 2 * omega;
    double* xs0 = dyn;
    double* xs1 = dyn + len;
    double* tw = dyn + 2 * len;
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < len; i += blockDim.x) { xs0[i] = 0.0; xs1[i] = 0.0; }
    for (int i = threadIdx.x; i < 2 * omega + 1; i += blockDim.x) tw[i] = taps[i];
    __syncthreads();
    for (int t = threadIdx.x; t < F; t += blockDim.x) xs0[omega + t] = b[(size_t)t * S + s];
    __syncthreads();
    double* cur = xs0;
    double* nxt = xs1;
    for (int it = 0; it < iters; ++it) {
        for (int t = threadIdx.x; t < F; t += blockDim.x) {
            double acc = 0.0;
            for (int d = 0; d <= 2 * omega; ++d) acc = __builtin_fma(tw[d], cur[t + d], acc);
            nxt[omega + t] = inv_on[t] * __builtin_fma(2.0 * lam[t], acc, b[(size_t)t * S + s]);
        }
        __syncthreads();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
    for (int t = threadIdx.x; t < F; t += blockDim.x) x_out[(size_t)t * S + s] = cur[omega + t];
}

template <int OMEGA, int K, int WAVES>
static int launch_wave(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                       int F, int S, int iters, hipStream_t st)
{
    hipLaunchKernelGGL((jacobi_wave_kernel<OMEGA, K, WAVES>), dim3(S), dim3(64 * WAVES), 0, st, b, x, taps, lam, inv_on, F, S,
                       iters);
    return hip_fail(hipGetLastError(), "jacobi_wave_kernel launch");
}

int launch_jacobi(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                  int F, int S, int omega, int iters, hipStream_t st)
{
    if (F <= 0 || S <= 0 || omega <= 0 || iters < 0) {
        set_error("mf_jacobi_f64: bad sizes F=%d S=%d omega=%d iters=%d", F, S, omega, iters);
        return MF_ERR_INVALID_ARG;
    }
    // the fewest frames per thread that covers the clip: a long clip spreads each series over up to 8 wavefronts
    // (measured faster than more frames per lane even when there are more series than SIMDs: tools/time_jacobi.py)
#define MF_JACOBI(O, K, WV) return launch_wave<O, K, WV>(b, x, taps, lam, inv_on, F, S, iters, st)
    // Far fewer series than SIMDs (1024): one wavefront per series leaves most SIMDs idle -- split each series over more
    // wavefronts with fewer frames per lane.  Measured (F = 300, 100 sweeps): 162 series (8 x 8 mesh) 40.7 us with one wavefront
    // per series, 31.8 with two, 30.4 with four; 578 series (16 x 16) 48.1 / 49.1 / 50.4 -- there the workgroup barrier of
    // a split series costs what the shorter per-lane loop saves, so the split starts below 512 series.
    static const int split = [] { const char* v = getenv("MF_JACOBI_SPLIT"); return v && *v ? atoi(v) : -1; }();   // tuning aid
    const int want = split > 0 ? split : (S < 256 ? 4 : S < 512 ? 2 : 1);
    if (omega == 10 && want > 1) {
        if (F <= 256 * 2 && want >= 4) MF_JACOBI(10, 2, 4);
        if (F <= 128 * 3) MF_JACOBI(10, 3, 2);
    }
    if (omega == 10) {
        if (F <= 64 * 5) MF_JACOBI(10, 5, 1);
        if (F <= 128 * 5) MF_JACOBI(10, 5, 2);
        if (F <= 256 * 5) MF_JACOBI(10, 5, 4);
        if (F <= 512 * 5) MF_JACOBI(10, 5, 8);
        if (F <= 512 * 10) MF_JACOBI(10, 10, 8);
        if (F <= 512 * 19) MF_JACOBI(10, 19, 8);
    } else if (omega == 5) {
        if (F <= 64 * 5) MF_JACOBI(5, 5, 1);
        if (F <= 128 * 5) MF_JACOBI(5, 5, 2);
        if (F <= 256 * 5) MF_JACOBI(5, 5, 4);
        if (F <= 512 * 5) MF_JACOBI(5, 5, 8);
    } else if (omega == 15) {
        if (F <= 64 * 8) MF_JACOBI(15, 8, 1);
        if (F <= 128 * 8) MF_JACOBI(15, 8, 2);
        if (F <= 256 * 8) MF_JACOBI(15, 8, 4);
        if (F <= 512 * 8) MF_JACOBI(15, 8, 8);
    } else if (omega == 20) {
        if (F <= 64 * 8) MF_JACOBI(20, 8, 1);
        if (F <= 128 * 8) MF_JACOBI(20, 8, 2);
        if (F <= 256 * 8) MF_JACOBI(20, 8, 4);
        if (F <= 512 * 8) MF_JACOBI(20, 8, 8);
    } else if (omega == 30) {
        if (F <= 64 * 10) MF_JACOBI(30, 10, 1);
        if (F <= 128 * 10) MF_JACOBI(30, 10, 2);
        if (F <= 256 * 10) MF_JACOBI(30, 10, 4);
        if (F <= 512 * 10) MF_JACOBI(30, 10, 8);
        if (F <= 512 * 19) MF_JACOBI(30, 19, 8);
    }
#undef MF_JACOBI
    const size_t lds = ((size_t)2 * (F + 2 * omega) + 2 * omega + 1) * sizeof(double);
    if (lds > 160 * 1024) {
        set_error("mf_jacobi_f64: F=%d omega=%d needs %zu bytes of LDS (> 160 KiB)", F, omega, lds);
        return MF_ERR_INVALID_ARG;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)jacobi_generic_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(jacobi_generic_kernel)");
    }
    hipLaunchKernelGGL(jacobi_generic_kernel, dim3(S), dim3(256), lds, st, b, x, taps, lam, inv_on, F, S, omega, iters);
    return hip_fail(hipGetLastError(), "jacobi_generic_kernel launch");
}

}  // namespace mf

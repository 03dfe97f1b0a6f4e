// The specialised Jacobi kernel (compile-time radius) shared by jacobi.hip and the ahead-of-time instantiation table
// jacobi_spec.hip.  See jacobi.hip for the mapping.
#pragma once
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

template <int OMEGA, int K, int WAVES>
static inline int launch_wave(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                       int F, int S, int iters, hipStream_t st)
{
    hipLaunchKernelGGL((jacobi_wave_kernel<OMEGA, K, WAVES>), dim3(S), dim3(64 * WAVES), 0, st, b, x, taps, lam, inv_on, F, S,
                       iters);
    return hip_fail(hipGetLastError(), "jacobi_wave_kernel launch");
}


// jacobi_spec.hip, compiled once per group of radii (MF_JACOBI_GROUP = 0..3, radii 8 g + 1 .. 8 g + 8): launches the
// kernel specialised for `omega` and the clip length, or returns MF_JACOBI_NOT_HERE when omega / F is not in its table.
#define MF_JACOBI_NOT_HERE 1
int launch_jacobi_spec_g0(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);
int launch_jacobi_spec_g1(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);
int launch_jacobi_spec_g2(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);
int launch_jacobi_spec_g3(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);

}  // namespace mf

// Stability score on the device (reference: meshflowstabilizer.py:1216-1259).
//
// Per vertex path (series) x[0..F-1]: profile p = diff(x) (N = F-1 samples), score = (energy of DFT bins 1..5) /
// (energy of all N bins).  The reference forms the whole spectrum with np.fft.fft; here the five bins are direct
// sums p_t * exp(-2 pi i k t / N) and the total comes from Parseval, sum_k |P_k|^2 = N * sum_t p_t^2 -- the same
// quantities up to float64 rounding (the host restatement in meshflow_amd/host.py stays the bit-for-bit one).
// The clip-level score is (mean over the x series + mean over the y series) / 2 (mfs.py:1256-1259).
//
// One wavefront per series, lanes stride over time; the per-series ratios are written out and summed by a
// single workgroup in a fixed order, so the result is deterministic.
#include "mf_common.h"

namespace mf {

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(64) void stability_series_kernel(const double* __restrict__ x, int F, int S, double* __restrict__ ratio)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const int N = F - 1;
    double tot = 0.0, re[5] = {0, 0, 0, 0, 0}, im[5] = {0, 0, 0, 0, 0};
    for (int t = lane; t < N; t += 64) {
        const double p = x[(size_t)(t + 1) * S + s] - x[(size_t)t * S + s];
        tot += p * p;
#pragma unroll
        for (int k = 1; k <= 5; ++k) {
            // angle 2 pi k t / N, reduced exactly in integers first: (k t) mod N
            const long long kt = ((long long)k * t) % N;
            double sn, cs;
            sincospi(2.0 * (double)kt / (double)N, &sn, &cs);
            re[k - 1] += p * cs;
            im[k - 1] -= p * sn;
        }
    }
    tot = wave_sum(tot);
    double low = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double r = wave_sum(re[k]), i = wave_sum(im[k]);
        if (k + 1 <= N - 1) low += r * r + i * i;              // the reference's slice [1:6] of N bins holds bins 1..min(5, N-1)
    }
    if (lane == 0) ratio[s] = low / ((double)N * tot);
}

__global__ __launch_bounds__(256) void stability_reduce_kernel(const double* __restrict__ ratio, int S, double* __restrict__ score)
{
    __shared__ double part[2][256];
    double sx = 0.0, sy = 0.0;
    for (int s = threadIdx.x; s < S; s += 256) {
        if (s & 1) sy += ratio[s]; else sx += ratio[s];        // series order: ..., vertex v x, vertex v y, ...
    }
    part[0][threadIdx.x] = sx;
    part[1][threadIdx.x] = sy;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            part[0][threadIdx.x] += part[0][threadIdx.x + w];
            part[1][threadIdx.x] += part[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double half = (double)(S / 2);
        score[0] = (part[0][0] / half + part[1][0] / half) / 2.0;
    }
}

}  // namespace

int launch_stability_score(const double* stab, int F, int S, double* ratio, double* score, hipStream_t st)
{
    // F = 1 has an empty velocity profile (np.fft.fft of it raises ValueError in the reference, mfs.py:1245); from F = 2 on
    // the slice [1:6] simply holds fewer bins (mfs.py:1250-1251)
    if (F < 2 || S < 2 || (S & 1)) {
        set_error("mf_stability_score_f64: needs at least 2 frames and an even number of series (F=%d S=%d)", F, S);
        return MF_ERR_INVALID_ARG;
    }
    stability_series_kernel<<<S, 64, 0, st>>>(stab, F, S, ratio);
    MF_HIP_TRY(hipGetLastError());
    stability_reduce_kernel<<<1, 256, 0, st>>>(ratio, S, score);
    MF_HIP_TRY(hipGetLastError());
    return MF_OK;
}

}  // namespace mf

// The specialised Jacobi kernel (compile-time radius) shared by jacobi.hip and the ahead-of-time instantiation table
// jacobi_spec.hip.  See jacobi.hip for the mapping.
#pragma once
#include <type_traits>

#include "mf_common.h"

namespace mf {

// The N sweeps of one thread's K frames.  SYM: the taps are symmetric (taps[OMEGA + d] == taps[OMEGA - d] bit for bit, as the
// reference builds them: exp(-((3 / OMEGA) (t - r))^2), mfs.py:750-752) and `w` holds the OMEGA + 1 distinct values w[|d|] -- half the
// scalar registers; the sum still runs over d = -OMEGA .. OMEGA in ascending order, so the bits are the same either way.
template <int OMEGA, int K, int WAVES, bool SYM>
__device__ __forceinline__ void jacobi_sweeps(double (&xs)[2][64 * WAVES * K + 2 * OMEGA], const double (&w)[SYM ? OMEGA + 1 : 2 * OMEGA + 1],
                                              const double (&bt)[K], const double (&two_lam)[K], const double (&inv)[K], double (&xn)[K],
                                              int lane, int iters)
{
    constexpr int NT = 2 * OMEGA + 1;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        double win[K + 2 * OMEGA];
        const double* src = &xs[cur][lane * K];
#pragma unroll
        for (int j = 0; j < K + 2 * OMEGA; ++j) win[j] = src[j];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int d = 0; d < NT; ++d) acc = __builtin_fma(w[SYM ? (d < OMEGA ? OMEGA - d : d - OMEGA) : d], win[k + d], acc);
            xn[k] = inv[k] * __builtin_fma(two_lam[k], acc, bt[k]);
        }
        double* dst = &xs[cur ^ 1][OMEGA + lane * K];
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = xn[k];
        cur ^= 1;
        __syncthreads();
    }
}

// PIPELINED form of the sweeps (K even): the K + 2 OMEGA window entries come from LDS as 16-byte pairs, a few pairs ahead of the FMAs that
// consume them, ONE read after every pair's 2 K FMAs -- instead of the whole window, a wait, then all K (2 OMEGA + 1) FMAs: with every
// wavefront of a CU in the same phase, the LDS pipe (8 wavefronts x 35 KB per sweep at OMEGA = 30) and the vector ALUs took turns.
// Entry j feeds output k through tap d = j - k, entries in ascending order: per output the taps still come in ascending d -- same bits.
// The loads are inline asm (the compiler would gather them in front of one wait again); what orders them against the FMAs is data
// flow: a wait "produces" the pair it makes valid, and a load is tied to an accumulator the FMAs in front of it have just written.
typedef double jd2_t __attribute__((ext_vector_type(2)));
template <int I, int N, typename Fn>
__device__ __forceinline__ void jacobi_static_for(Fn&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        jacobi_static_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ void jacobi_lds_pair(jd2_t& dst, uint32_t addr, double& tie)
{
    asm volatile("ds_read_b128 %0, %2 offset:%3" : "=v"(dst), "+v"(tie) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void jacobi_lds_wait(jd2_t& v)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
#ifndef MF_JACOBI_AHEAD
#define MF_JACOBI_AHEAD 4
#endif
template <int OMEGA, int K, int WAVES, bool SYM>
__device__ __forceinline__ void jacobi_sweeps_piped(double (&xs)[2][64 * WAVES * K + 2 * OMEGA], const double (&w)[SYM ? OMEGA + 1 : 2 * OMEGA + 1],
                                                    const double (&bt)[K], const double (&two_lam)[K], const double (&inv)[K], double (&xn)[K],
                                                    int lane, int iters)
{
    static_assert(K % 2 == 0, "pairs of window entries");
    constexpr int NT = 2 * OMEGA + 1, NP = (K + 2 * OMEGA) / 2, A = MF_JACOBI_AHEAD < NP ? MF_JACOBI_AHEAD : NP;
    const uint32_t base[2] = { (uint32_t)(uintptr_t)&xs[0][lane * K], (uint32_t)(uintptr_t)&xs[1][lane * K] };
    // (the taps are in their scalar registers BEFORE the loop: otherwise the compiler, which cannot see the asm loads, waits for "its"
    // scalar loads with lgkmcnt(0) inside the loop, behind the first window loads of every sweep)
#pragma unroll
    for (int d = 0; d < (SYM ? OMEGA + 1 : 2 * OMEGA + 1); ++d) asm volatile("" :: "s"(w[d]));
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t addr = base[cur];
        jd2_t wv[NP];
        double acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.0;
        jacobi_static_for<0, A>([&](auto pc) { constexpr int P = decltype(pc)::value; jacobi_lds_pair<16 * P>(wv[P], addr, acc[0]); });
        jacobi_static_for<0, NP>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            jacobi_lds_wait<(A - 1 < NP - 1 - P ? A - 1 : NP - 1 - P)>(wv[P]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int d = 2 * P + e - k;
                    if (d >= 0 && d < NT) acc[k] = __builtin_fma(w[SYM ? (d < OMEGA ? OMEGA - d : d - OMEGA) : d], wv[P][e], acc[k]);
                }
            if constexpr (P + A < NP) jacobi_lds_pair<16 * (P + A)>(wv[P + A], addr, acc[2 * P + 1 < K - 1 ? 2 * P + 1 : K - 1]);
        });
#pragma unroll
        for (int k = 0; k < K; ++k) xn[k] = inv[k] * __builtin_fma(two_lam[k], acc[k], bt[k]);
        double* dst = &xs[cur ^ 1][OMEGA + lane * K];
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = xn[k];
        cur ^= 1;
        __syncthreads();
    }
}

// A clip too long for LDS (F > 64 WAVES K) is swept in TIME TILES (jacobi.hip, launch_jacobi_tiled): a workgroup holds frames
// [j T - halo, j T + T + halo) of one series, runs `iters` <= halo / OMEGA sweeps on them from x_in and writes frames [j T, j T + T) --
// after k sweeps a frame's value depends on the frames within k OMEGA of it only (x_new[t] reads x[t - OMEGA .. t + OMEGA]), so what
// the halo rows lack of their own neighbourhood never reaches the rows that are written; frames outside the clip are the zero halo
// of the untiled kernels.  Same operations per frame in the same order: same bits.
struct JacobiTile { const double* x_in; int T, halo; };

// The body of the specialised kernel.  TILED: blockIdx.y = time tile (above); otherwise the whole clip from x = b.
template <int OMEGA, int K, int WAVES, bool TILED>
__device__ __forceinline__ void jacobi_wave_body(const double* __restrict__ b, double* __restrict__ x_out,
                                                 const double* __restrict__ taps,
                                                 const double* __restrict__ lam,
                                                 const double* __restrict__ inv_on, int F, int S, int iters, int s0, const JacobiTile& tile)
{
    constexpr int NT = 2 * OMEGA + 1;
    constexpr int NTHR = 64 * WAVES;
    constexpr int LEN = NTHR * K + 2 * OMEGA;
    // Beyond ~45 taps the 2 NT scalar registers of the float64 taps do not fit next to everything else (OMEGA = 30: 122 of them,
    // 110 spilled to vector-register lanes and restored every sweep); symmetric taps -- what the reference always has -- need
    // OMEGA + 1 values.  Decided per launch by a wave-uniform bit comparison; anything else takes the full table.
    constexpr bool TRY_SYM = NT > 45;
    __shared__ __attribute__((aligned(16))) double xs[2][LEN];      // (16 bytes: the pipelined sweeps read it in ds_read_b128 pairs)
    const int s = s0 + (int)blockIdx.x;          // (series s0 .. s0 + gridDim.x - 1 of the S: a launch may cover a slice)
    const int lane = threadIdx.x;

    // frames this thread owns: t0 .. t0 + K - 1; frames [w_lo, w_hi) are written
    int t0 = lane * K, w_lo = 0, w_hi = F, r_hi = F;
    if constexpr (TILED) {
        w_lo = (int)blockIdx.y * tile.T;
        w_hi = w_lo + tile.T < F ? w_lo + tile.T : F;
        r_hi = w_hi + tile.halo < F ? w_hi + tile.halo : F;          // (rows beyond the halo are never needed: left at zero)
        t0 += w_lo - tile.halo;
    }
    double bt[K], two_lam[K], inv[K], xn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = t0 + k;
        const bool in = TILED ? (t >= 0 && t < r_hi) : t < F;
        bt[k] = in ? b[(size_t)t * S + s] : 0.0;
        two_lam[k] = in ? 2.0 * lam[t] : 0.0;
        inv[k] = in ? inv_on[t] : 0.0;            // frames past the end stay exactly 0 = the zero halo
        xn[k] = bt[k];                            // x_start = b (mfs.py:871) ...
        if constexpr (TILED)
            if (tile.x_in != b) xn[k] = in ? tile.x_in[(size_t)t * S + s] : 0.0;     // ... or the state the sweeps before this launch left
    }
    for (int i = lane; i < LEN; i += NTHR) { xs[0][i] = 0.0; xs[1][i] = 0.0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) xs[0][OMEGA + lane * K + k] = xn[k];
    __syncthreads();

    bool symmetric = false;
    if (TRY_SYM) {
        typedef const __attribute__((address_space(4))) unsigned long long* cbits_t;      // (scalar loads)
        const cbits_t tb = (cbits_t)(uintptr_t)taps;
        unsigned long long diff = 0;
#pragma unroll
        for (int d = 1; d <= OMEGA; ++d) diff |= tb[OMEGA + d] ^ tb[OMEGA - d];
        symmetric = diff == 0;
    }
    if (TRY_SYM && symmetric) {
        double w[OMEGA + 1];
#pragma unroll
        for (int d = 0; d <= OMEGA; ++d) w[d] = taps[OMEGA + d];
        if constexpr (K % 2 == 0 && K >= 8 && WAVES == 1) jacobi_sweeps_piped<OMEGA, K, WAVES, true>(xs, w, bt, two_lam, inv, xn, lane, iters);
        else jacobi_sweeps<OMEGA, K, WAVES, true>(xs, w, bt, two_lam, inv, xn, lane, iters);
    } else if (TRY_SYM) {
        // Asymmetric taps at a wide radius (never from the reference; the C ABI accepts any): the full table does not fit the scalar
        // registers, and carrying it would cost the symmetric path its occupancy -- so a compact loop with the taps in LDS (wave-uniform
        // reads), same order of operations, a few times slower.
        __shared__ double s_taps[NT];
        for (int d = lane; d < NT; d += NTHR) s_taps[d] = taps[d];
        __syncthreads();
        int cur = 0;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const double* src = &xs[cur][lane * K + k];
                double acc = 0.0;
#pragma unroll 1
                for (int d = 0; d < NT; ++d) acc = __builtin_fma(s_taps[d], src[d], acc);
                xn[k] = inv[k] * __builtin_fma(two_lam[k], acc, bt[k]);
            }
            double* dst = &xs[cur ^ 1][OMEGA + lane * K];
#pragma unroll
            for (int k = 0; k < K; ++k) dst[k] = xn[k];
            cur ^= 1;
            __syncthreads();
        }
    } else {
        double w[NT];
#pragma unroll
        for (int d = 0; d < NT; ++d) w[d] = taps[d];
        if constexpr (K % 2 == 0 && K >= 8 && WAVES == 1) jacobi_sweeps_piped<OMEGA, K, WAVES, false>(xs, w, bt, two_lam, inv, xn, lane, iters);
        else jacobi_sweeps<OMEGA, K, WAVES, false>(xs, w, bt, two_lam, inv, xn, lane, iters);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = t0 + k;
        if (TILED ? (t >= w_lo && t < w_hi) : t < F) x_out[(size_t)t * S + s] = xn[k];
    }
}

// (Series spread over several wavefronts at a wide radius keep two wavefronts per SIMD as their register budget -- what they had
// while the full tap table shared the kernel: scheduled for four, the barrier-coupled wavefronts measured 14 % slower.)
template <int OMEGA, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(1, (2 * OMEGA + 1 > 45 && WAVES > 1 && K >= 8) ? 2 : 8)))
void jacobi_wave_kernel(const double* __restrict__ b, double* __restrict__ x_out,
                                                         const double* __restrict__ taps,
                                                         const double* __restrict__ lam,
                                                         const double* __restrict__ inv_on, int F, int S, int iters, int s0)
{
    jacobi_wave_body<OMEGA, K, WAVES, false>(b, x_out, taps, lam, inv_on, F, S, iters, s0, JacobiTile{});
}

// grid = (S, time tiles)
template <int OMEGA, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(1, (2 * OMEGA + 1 > 45 && WAVES > 1 && K >= 8) ? 2 : 8)))
void jacobi_wave_tiled_kernel(const double* __restrict__ b, double* __restrict__ x_out,
                                                         const double* __restrict__ taps,
                                                         const double* __restrict__ lam,
                                                         const double* __restrict__ inv_on, int F, int S, int iters, JacobiTile tile)
{
    jacobi_wave_body<OMEGA, K, WAVES, true>(b, x_out, taps, lam, inv_on, F, S, iters, 0, tile);
}

template <int OMEGA, int K, int WAVES>
static inline int launch_wave_tiled(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                                    int F, int S, int iters, const JacobiTile& tile, int ntiles, hipStream_t st)
{
    hipLaunchKernelGGL((jacobi_wave_tiled_kernel<OMEGA, K, WAVES>), dim3(S, ntiles), dim3(64 * WAVES), 0, st, b, x, taps, lam, inv_on, F, S, iters, tile);
    return hip_fail(hipGetLastError(), "jacobi_wave_tiled_kernel launch");
}

template <int OMEGA, int K, int WAVES>
static inline int launch_wave(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                       int F, int S, int iters, hipStream_t st, int s0 = 0, int count = -1)
{
    hipLaunchKernelGGL((jacobi_wave_kernel<OMEGA, K, WAVES>), dim3(count < 0 ? S : count), dim3(64 * WAVES), 0, st, b, x, taps, lam, inv_on, F, S,
                       iters, s0);
    return hip_fail(hipGetLastError(), "jacobi_wave_kernel launch");
}

// jacobi.hip: a second stream of the current device for launches that should run BESIDE the ones on `st` (the tail of a sweep whose
// series do not fill the chip's SIMDs evenly).  fork: everything queued on `st` so far precedes what is then queued on *side; join:
// what was queued on the side stream precedes everything queued on `st` afterwards.  jacobi_simd_count: SIMDs of the current device.
int jacobi_side_fork(hipStream_t st, hipStream_t* side);
int jacobi_side_join(hipStream_t st);
int jacobi_simd_count();


// jacobi_spec.hip: `iters` sweeps of one tiled launch (frames per workgroup: MF_JACOBI_TILE_LEN) by the kernel specialised for
// `omega`, or MF_JACOBI_NOT_HERE when that radius has no tiled specialisation (jacobi.hip then takes the run-time-radius form).
#define MF_JACOBI_TILE_K 19
#define MF_JACOBI_TILE_WAVES 8
#define MF_JACOBI_TILE_LEN (64 * MF_JACOBI_TILE_WAVES * MF_JACOBI_TILE_K)
int launch_jacobi_tiled_spec(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega,
                             int iters, const JacobiTile& tile, int ntiles, hipStream_t st);
// Frames a tile of the specialised kernel for `omega` holds (0: none -- the run-time-radius form with MF_JACOBI_TILE_LEN frames takes it).
int jacobi_tiled_spec_len(int omega);

// jacobi_spec.hip, compiled once per group of radii (MF_JACOBI_GROUP = 0..3, radii 8 g + 1 .. 8 g + 8): launches the
// kernel specialised for `omega` and the clip length, or returns MF_JACOBI_NOT_HERE when omega / F is not in its table.
#define MF_JACOBI_NOT_HERE 1
int launch_jacobi_spec_g0(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);
int launch_jacobi_spec_g1(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);
int launch_jacobi_spec_g2(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);
int launch_jacobi_spec_g3(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega, int iters, int want, hipStream_t st);

}  // namespace mf

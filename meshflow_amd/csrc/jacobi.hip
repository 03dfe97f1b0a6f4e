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
// One wavefront per series with K >= 8 (radii 13..32) does not hold the whole window: its entries arrive as 16-byte pairs a few pairs
// ahead of the FMAs that consume them (jacobi_kernels.h, jacobi_sweeps_piped), so the LDS pipe and the vector ALUs work side by side.
#include <stdlib.h>

#include <mutex>

#include "jacobi_kernels.h"

namespace mf {

// A second stream per device for launches that run beside the caller's stream (jacobi_kernels.h), with the two events of the fork /
// join.  Created on first use, kept for the life of the process (one small object per device).
namespace {
struct SideStream { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; int simds = 0; int state = 0; std::mutex busy; };
SideStream g_side[64];
std::mutex g_side_lock;
SideStream* side_for_current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(g_side_lock);
    SideStream& d = g_side[dev];
    if (d.state == 0) {
        int cus = 0;
        d.state = -1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&d.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&d.join, hipEventDisableTiming) == hipSuccess) {
            d.simds = 4 * cus;
            d.state = 1;
        }
    }
    return d.state == 1 ? &d : nullptr;
}
}  // namespace

int jacobi_simd_count()
{
    SideStream* d = side_for_current_device();
    return d ? d->simds : 0;
}
int jacobi_side_fork(hipStream_t st, hipStream_t* side)
{
    SideStream* d = side_for_current_device();
    if (!d) return MF_ERR_HIP;
    d->busy.lock();                       // (the event pair is per device: one fork ... join at a time; released by jacobi_side_join)
    hipError_t e = hipEventRecord(d->fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(d->s, d->fork, 0);
    if (e != hipSuccess) { d->busy.unlock(); return hip_fail(e, "jacobi side stream fork"); }
    *side = d->s;
    return MF_OK;
}
int jacobi_side_join(hipStream_t st)
{
    SideStream* d = side_for_current_device();
    if (!d) return MF_ERR_HIP;
    hipError_t e = hipEventRecord(d->join, d->s);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, d->join, 0);
    d->busy.unlock();
    return hip_fail(e, "jacobi side stream join");
}

// Any other radius (mfs.py:46 accepts every temporal_smoothing_radius): the same mapping with the radius at RUN time.  A thread
// still owns K consecutive frames and all K accumulators, but instead of holding its whole K + 2 omega window in registers it
// slides a K-entry window over it: tap d multiplies the window, then the entry that tap d+1 no longer needs is replaced by the
// next one from LDS (one ds_read_b64 per tap; unrolled by K, so the rotation is a renaming).  The taps are wave-uniform scalar
// loads, K at a time (constant address space).  Every output sums its taps in ascending order from zero, exactly like the
// specialised kernel and the CPU oracle -- identical bits.  K is odd: a lane stride of K doubles keeps the 64-bit LDS reads
// free of bank conflicts.  ~40 VGPRs whatever the radius.
typedef const __attribute__((address_space(4))) double* ctap_t;

// TILED (clips beyond 64 WAVES K frames): blockIdx.y = time tile, `iters` sweeps from tile.x_in -- jacobi_kernels.h, JacobiTile.
template <int K, int WAVES, bool TILED>
__device__ __forceinline__ void jacobi_runtime_body(const double* __restrict__ b, double* __restrict__ x_out,
                                                    const double* __restrict__ taps_g, const double* __restrict__ lam,
                                                    const double* __restrict__ inv_on, int F, int S, int omega, int iters, const JacobiTile& tile)
{
    extern __shared__ double dyn[];
    constexpr int NTHR = 64 * WAVES;
    const int NT = 2 * omega + 1;
    const int LEN = NTHR * K + 2 * omega + K;      // + K: the sliding window runs K - 1 entries past the last tap's reach
    double* xs0 = dyn;
    double* xs1 = dyn + LEN;
    const ctap_t taps = (ctap_t)(uintptr_t)taps_g;
    const int s = blockIdx.x;
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
    for (int i = lane; i < LEN; i += NTHR) { xs0[i] = 0.0; xs1[i] = 0.0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) xs0[omega + lane * K + k] = xn[k];
    __syncthreads();

    double* cur = xs0;
    double* nxt = xs1;
    for (int it = 0; it < iters; ++it) {
        const double* src = cur + lane * K;        // src[j] = x[frame lane K + j - omega]
        double r[K], acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { r[k] = src[k]; acc[k] = 0.0; }
        // K taps per scalar load.  (Measured and rejected: fetching the next K under this block's FMAs, +20 %; all taps of a small
        // radius loaded once and kept in scalar registers for every sweep, +10-30 % -- scalar-register spills; taps spread over the
        // lanes of a register and fetched by v_readlane, 2x.)
        int d = 0;
        for (; d + K <= NT; d += K) {
            double w[K];
#pragma unroll
            for (int j = 0; j < K; ++j) w[j] = taps[d + j];
#pragma unroll
            for (int j = 0; j < K; ++j) {
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = __builtin_fma(w[j], r[(j + k) % K], acc[k]);
                r[j] = src[d + j + K];             // slot j held x[.. + d + j]: tap d + j + 1 starts one entry later
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j)                // the last NT - d < K taps (wave-uniform tests)
            if (d + j < NT) {
                const double w = taps[d + j];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = __builtin_fma(w, r[(j + k) % K], acc[k]);
                r[j] = src[d + j + K];
            }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            xn[k] = inv[k] * __builtin_fma(two_lam[k], acc[k], bt[k]);
            nxt[omega + lane * K + k] = xn[k];
        }
        double* tmp = cur; cur = nxt; nxt = tmp;
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int t = t0 + k;
        if (TILED ? (t >= w_lo && t < w_hi) : t < F) x_out[(size_t)t * S + s] = xn[k];
    }
}

template <int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void jacobi_runtime_kernel(const double* __restrict__ b, double* __restrict__ x_out,
                                                                    const double* __restrict__ taps_g, const double* __restrict__ lam,
                                                                    const double* __restrict__ inv_on, int F, int S, int omega, int iters)
{
    jacobi_runtime_body<K, WAVES, false>(b, x_out, taps_g, lam, inv_on, F, S, omega, iters, JacobiTile{});
}

// grid = (S, time tiles)
template <int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void jacobi_runtime_tiled_kernel(const double* __restrict__ b, double* __restrict__ x_out,
                                                                          const double* __restrict__ taps_g, const double* __restrict__ lam,
                                                                          const double* __restrict__ inv_on, int F, int S, int omega, int iters,
                                                                          JacobiTile tile)
{
    jacobi_runtime_body<K, WAVES, true>(b, x_out, taps_g, lam, inv_on, F, S, omega, iters, tile);
}

template <int K, int WAVES>
static int launch_runtime(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                          int F, int S, int omega, int iters, hipStream_t st)
{
    const size_t lds = (size_t)2 * (64 * WAVES * K + 2 * omega + K) * sizeof(double);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)jacobi_runtime_kernel<K, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(jacobi_runtime_kernel)");
    }
    hipLaunchKernelGGL((jacobi_runtime_kernel<K, WAVES>), dim3(S), dim3(64 * WAVES), lds, st, b, x, taps, lam, inv_on, F, S, omega, iters);
    return hip_fail(hipGetLastError(), "jacobi_runtime_kernel launch");
}

// Last resort (a clip too long for the kernels above) -- one 256-thread workgroup per series,
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

// ---- clips of ANY length (mfs.py:193-213 reads every frame of the file; mfs.py:632-710, 871-878 work for any num_frames) ----------
//
// The kernels above keep a series' whole time axis in LDS: 64 x 8 x 19 = 9,728 frames at most.  Beyond that the clip is swept in TIME
// TILES: a workgroup takes MF_JACOBI_TILE_LEN consecutive frames of one series -- T frames it will write plus a halo of ks * omega frames
// either side -- and runs ks sweeps on them in LDS exactly as the untiled kernels do; a frame's value after ks sweeps depends on the
// frames within ks * omega of it only, so the T inner frames come out as the whole-clip sweep computes them, bit for bit.  All `iters`
// sweeps take ceil(iters / ks_max) launches that ping-pong between x and one scratch array of F x S doubles (stream-ordered
// allocation); b is only ever read.  ks_max keeps the halo at a quarter of the tile or less (at omega = 10 all 100 sweeps of the
// default configuration are ONE launch: halo 1,000, T = 7,728 -- 26 % redundant rows).  HBM/L2 traffic: (x_in + b in, x out) per LAUNCH,
// not per sweep.
struct ScratchAsync {
    void* p = nullptr;
    hipStream_t st = nullptr;
    hipError_t alloc(size_t bytes, hipStream_t s) { st = s; return hipMallocAsync(&p, bytes, s); }
    ~ScratchAsync() { if (p) (void)hipFreeAsync(p, st); }
};

template <typename Launch>
static int sweep_launches(const double* b, double* x, int F, int S, int iters, int ks_max, hipStream_t st, Launch&& launch)
{
    if (iters == 0) {                                   // x = x_start = b (mfs.py:871)
        MF_HIP_TRY(hipMemcpyAsync(x, b, (size_t)F * S * sizeof(double), hipMemcpyDeviceToDevice, st));
        return MF_OK;
    }
    const int launches = (iters + ks_max - 1) / ks_max;
    ScratchAsync tmp;
    if (launches > 1) MF_HIP_TRY(tmp.alloc((size_t)F * S * sizeof(double), st));
    const double* src = b;
    int done = 0;
    for (int i = 0; i < launches; ++i) {
        const int ks = (iters - done + (launches - i) - 1) / (launches - i);          // the sweeps spread evenly over the launches
        double* dst = ((launches - 1 - i) % 2 == 0) ? x : (double*)tmp.p;             // the last launch writes x
        if (const int rc = launch(src, dst, ks)) return rc;
        src = dst;
        done += ks;
    }
    return MF_OK;
}

static int launch_jacobi_tiled(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                               int F, int S, int omega, int iters, hipStream_t st)
{
    constexpr int K = MF_JACOBI_TILE_K, WAVES = MF_JACOBI_TILE_WAVES;
    const char* fr = getenv("MF_JACOBI_RUNTIME");                     // testing aid, read at every call: the run-time-radius form
    const bool force_runtime = fr && *fr == '1';
    const int spec_len = force_runtime ? 0 : jacobi_tiled_spec_len(omega);
    const int LEN = spec_len > 0 ? spec_len : MF_JACOBI_TILE_LEN;       // frames a tile holds: the specialised kernel's, or the run-time-radius form's
    const int ks_max = (LEN / 4) / omega > 1 ? (LEN / 4) / omega : 1;
    return sweep_launches(b, x, F, S, iters, ks_max, st, [&](const double* src, double* dst, int ks) {
        JacobiTile tile;
        tile.x_in = src;
        tile.halo = ks * omega;
        int t_max = LEN - 2 * tile.halo;                              // >= LEN / 2
        if (const char* v = getenv("MF_JACOBI_TILE_T")) { const int cap = atoi(v); if (cap > 0 && cap < t_max) t_max = cap; }   // testing aid
        const int ntiles = (F + t_max - 1) / t_max;
        tile.T = (F + ntiles - 1) / ntiles;                           // (evened out: every tile costs the same whatever it writes)
        if (ntiles > 65535) { set_error("mf_jacobi_f64: F=%d needs %d time tiles (> 65535)", F, ntiles); return (int)MF_ERR_INVALID_ARG; }
        if (spec_len > 0) {
            const int rc = launch_jacobi_tiled_spec(b, dst, taps, lam, inv_on, F, S, omega, ks, tile, ntiles, st);
            if (rc != MF_JACOBI_NOT_HERE) return rc;
        }
        const size_t lds = (size_t)2 * (LEN + 2 * omega + K) * sizeof(double);
        hipError_t e = hipFuncSetAttribute((const void*)jacobi_runtime_tiled_kernel<K, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(jacobi_runtime_tiled_kernel)");
        hipLaunchKernelGGL((jacobi_runtime_tiled_kernel<K, WAVES>), dim3(S, ntiles), dim3(64 * WAVES), lds, st, b, dst, taps, lam, inv_on, F, S, omega, ks, tile);
        return hip_fail(hipGetLastError(), "jacobi_runtime_tiled_kernel launch");
    });
}
static bool jacobi_tiled_fits(int omega) { return (size_t)2 * (MF_JACOBI_TILE_LEN + 2 * omega + MF_JACOBI_TILE_K) * sizeof(double) <= 160 * 1024; }   // omega <= 246

// Whatever is left -- a radius of hundreds of frames on a clip too long for LDS: one launch per sweep on the arrays in global memory
// (ping-pong between x and a scratch array), lanes along the series so that every tap reads one contiguous run of the [F][S] layout;
// taps in ascending order from zero, one fma each, like every other form: same bits.  No shape is refused.
__global__ __launch_bounds__(256) void jacobi_global_sweep_kernel(const double* __restrict__ b, const double* __restrict__ x_in, double* __restrict__ x_out,
                                                                  const double* __restrict__ taps, const double* __restrict__ lam,
                                                                  const double* __restrict__ inv_on, int F, int S, int omega, unsigned s_blocks)
{
    const unsigned tb = blockIdx.x / s_blocks, sb = blockIdx.x - tb * s_blocks;
    const int s = (int)(sb * 64u + (threadIdx.x & 63u));
    const long long t = (long long)tb * 4 + (threadIdx.x >> 6);
    if (s >= S || t >= F) return;
    double acc = 0.0;
    for (int d = 0; d <= 2 * omega; ++d) {
        const long long r = t + d - omega;
        if (r >= 0 && r < F) acc = __builtin_fma(taps[d], x_in[(size_t)r * S + s], acc);
    }
    x_out[(size_t)t * S + s] = inv_on[t] * __builtin_fma(2.0 * lam[t], acc, b[(size_t)t * S + s]);
}

static int launch_jacobi_global(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                                int F, int S, int omega, int iters, hipStream_t st)
{
    const unsigned s_blocks = (unsigned)((S + 63) / 64);
    const unsigned long long blocks = (unsigned long long)s_blocks * (unsigned long long)((F + 3) / 4);
    if (blocks > 0x7FFFFFFFull) { set_error("mf_jacobi_f64: F=%d S=%d is beyond one launch grid", F, S); return MF_ERR_INVALID_ARG; }
    return sweep_launches(b, x, F, S, iters, 1, st, [&](const double* src, double* dst, int) {
        hipLaunchKernelGGL(jacobi_global_sweep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, b, src, dst, taps, lam, inv_on, F, S, omega, s_blocks);
        return hip_fail(hipGetLastError(), "jacobi_global_sweep_kernel launch");
    });
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
    // Far fewer series than SIMDs (1024): one wavefront per series leaves most SIMDs idle -- split each series over more
    // wavefronts with fewer frames per lane.  Measured (F = 300, 100 sweeps): 162 series (8 x 8 mesh) 40.7 us with one wavefront
    // per series, 31.8 with two, 30.4 with four; 578 series (16 x 16) 48.1 / 49.1 / 50.4 -- there the workgroup barrier of
    // a split series costs what the shorter per-lane loop saves, so the split starts below 512 series.
    // testing aid, read at every call: MF_JACOBI_LONG=1 sends any clip through the time tiles (MF_JACOBI_TILE_T caps the frames a tile
    // writes, so that short clips have seams too), =2 through the sweep-by-sweep form in global memory
    if (const char* v = getenv("MF_JACOBI_LONG")) {
        if (*v == '1' && jacobi_tiled_fits(omega)) return launch_jacobi_tiled(b, x, taps, lam, inv_on, F, S, omega, iters, st);
        if (*v == '2') return launch_jacobi_global(b, x, taps, lam, inv_on, F, S, omega, iters, st);
    }
    static const int split = [] { const char* v = getenv("MF_JACOBI_SPLIT"); return v && *v ? atoi(v) : -1; }();   // tuning aid
    const int want = split > 0 ? split : (S < 256 ? 4 : S < 512 ? 2 : 1);
    static const bool force_runtime = [] { const char* v = getenv("MF_JACOBI_RUNTIME"); return v && *v == '1'; }();   // tuning aid
    if (force_runtime) omega = -omega;                  // (skips the specialised radii below)
    // Radii 1..32 have kernels specialised ahead of time (jacobi_spec.hip: the radius, the frames per thread and the wavefronts
    // per series are compile-time constants -- whole window in registers, taps in scalar registers for all sweeps)
    if (omega >= 1 && omega <= 32) {
        int rc = MF_JACOBI_NOT_HERE;
        switch ((omega - 1) / 8) {
        case 0: rc = launch_jacobi_spec_g0(b, x, taps, lam, inv_on, F, S, omega, iters, want, st); break;
        case 1: rc = launch_jacobi_spec_g1(b, x, taps, lam, inv_on, F, S, omega, iters, want, st); break;
        case 2: rc = launch_jacobi_spec_g2(b, x, taps, lam, inv_on, F, S, omega, iters, want, st); break;
        default: rc = launch_jacobi_spec_g3(b, x, taps, lam, inv_on, F, S, omega, iters, want, st); break;
        }
        if (rc != MF_JACOBI_NOT_HERE) return rc;
    }
    // every other radius: the run-time-radius kernel.  Frames per thread K in {3, 5, 7} (19 for clips beyond 3584 frames), wavefronts per
    // series = what covers the clip.  These sweeps are latency-bound (a sweep is one dependent pass per wavefront), so the fewest
    // frames per thread win as long as the series' wavefronts still fit the chip about twice over (1024 SIMDs).
    if (omega < 0) omega = -omega;
    {
        const auto waves_for = [&](int k) { int wv = 1; while (wv < 8 && F > 64 * wv * k) wv *= 2; return wv; };
        const auto fits = [&](int k, int wv) { return F <= 64 * wv * k && (size_t)2 * (64 * wv * k + 2 * omega + k) * sizeof(double) <= 160 * 1024; };
#define MF_JACOBI_RT(K, WV) if (wv == WV) return launch_runtime<K, WV>(b, x, taps, lam, inv_on, F, S, omega, iters, st)
#define MF_JACOBI_RT_K(K) { MF_JACOBI_RT(K, 1); MF_JACOBI_RT(K, 2); MF_JACOBI_RT(K, 4); MF_JACOBI_RT(K, 8); }
        for (const int k : { 3, 5, 7 }) {
            const int wv = waves_for(k);
            if (!fits(k, wv) || (k != 7 && (long long)S * wv > 2048)) continue;
            if (k == 3) MF_JACOBI_RT_K(3) else if (k == 5) MF_JACOBI_RT_K(5) else MF_JACOBI_RT_K(7)
        }
        if (fits(19, 8)) return launch_runtime<19, 8>(b, x, taps, lam, inv_on, F, S, omega, iters, st);
#undef MF_JACOBI_RT_K
#undef MF_JACOBI_RT
    }
    // a clip beyond every LDS kernel: time tiles (any radius up to 246), else sweep by sweep in global memory
    if (F > MF_JACOBI_TILE_LEN && jacobi_tiled_fits(omega)) return launch_jacobi_tiled(b, x, taps, lam, inv_on, F, S, omega, iters, st);
    const size_t lds = ((size_t)2 * (F + 2 * omega) + 2 * omega + 1) * sizeof(double);
    if (lds > 160 * 1024) return launch_jacobi_global(b, x, taps, lam, inv_on, F, S, omega, iters, st);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)jacobi_generic_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(jacobi_generic_kernel)");
    }
    hipLaunchKernelGGL(jacobi_generic_kernel, dim3(S), dim3(256), lds, st, b, x, taps, lam, inv_on, F, S, omega, iters);
    return hip_fail(hipGetLastError(), "jacobi_generic_kernel launch");
}

}  // namespace mf

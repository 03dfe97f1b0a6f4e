// Ahead-of-time table of the specialised Jacobi kernels (jacobi_kernels.h): compiled once per group of eight radii
// (-DMF_JACOBI_GROUP=g: radii 8 g + 1 .. 8 g + 8), so that the groups build in parallel.  Per radius: the fewest frames per thread K
// (5 up to omega = 12, 8 up to 20, 10 beyond) x 1 / 2 / 4 / 8 wavefronts per series, i.e. clips of up
// to 512 K frames; omega = 10 and 30 (BASELINE configs 2-4) also have the long-clip variants (K = 10, 19).  Longer clips and larger
// radii take jacobi.hip's run-time-radius kernel.
#include <stdlib.h>

#include "jacobi_kernels.h"

#ifndef MF_JACOBI_GROUP
#error "compile with -DMF_JACOBI_GROUP=0..3"
#endif

namespace mf {

#define MF_CAT2(a, b) a##b
#define MF_CAT(a, b) MF_CAT2(a, b)

namespace {

template <int OMEGA>
int launch_radius(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int iters, int want, hipStream_t st)
{
    constexpr int K = OMEGA <= 12 ? 5 : OMEGA <= 20 ? 8 : 10;
#define MF_JACOBI(O, KK, WV) return launch_wave<O, KK, WV>(b, x, taps, lam, inv_on, F, S, iters, st)
    // Far fewer series than SIMDs (1024): one wavefront per series leaves most SIMDs idle -- split each series over more
    // wavefronts with fewer frames per lane (measured at omega = 10, F = 300, 100 sweeps: 162 series 40.7 us with one wavefront
    // per series, 31.8 with two, 30.4 with four; 578 series 48.1 / 49.1 / 50.4: the split starts below 512 series)
    if (OMEGA == 10 && want > 1) {
        if (F <= 256 * 2 && want >= 4) MF_JACOBI(10, 2, 4);
        if (F <= 128 * 3) MF_JACOBI(10, 3, 2);
    }
    // One wavefront per series; a SIMD's float64 pipe is full from two wavefronts on (F = 600, omega = 30: 1024 series 320 us, 2048 543,
    // 3072 789), so the sweep takes (series on the busiest SIMD) x (time of one series).  With S = q SIMDs + r
    // series, r SIMDs would carry one series more than the others (config 3: 2178 series on 1024 SIMDs).  A small remainder is therefore
    // cut into four wavefronts per series (3 frames per lane, a workgroup barrier per sweep) and launched BESIDE the main launch on a
    // second stream: 4 r short pieces spread over 4 r SIMDs instead of r long ones (the pipelined K = 10 kernel's 125 registers leave them
    // room beside two main wavefronts; at the 236 of the first version they had to wait for main wavefronts to retire).  Same arithmetic
    // per frame in the same order: same bits.
    if (OMEGA > 20 && want == 1 && F <= 64 * K && F <= 256 * 3) {
        static const bool no_tail = [] { const char* v = getenv("MF_JACOBI_NO_TAIL"); return v && *v == '1'; }();       // tuning aid
        const int simds = jacobi_simd_count(), r = simds > 0 ? S % simds : 0;
        hipStream_t side = nullptr;
        if (!no_tail && S > simds && r > 0 && 4 * r <= simds && jacobi_side_fork(st, &side) == MF_OK) {
            static const int tail_waves = [] { const char* v = getenv("MF_JACOBI_TAIL_WAVES"); return v && *v ? atoi(v) : 4; }();   // tuning aid
            // The PIECES go to the caller's stream and the main launch to the side stream: behind another kernel a launch on the caller's
            // stream starts at once and one on the side stream ~6 us later (a cross-queue event), and the pieces must be the OLDER wavefronts
            // of their SIMDs -- then they are done in 180 us; as the younger ones they crawl beside the main wavefronts for the whole
            // sweep and the sweep takes 200 us longer (rocprofv3 trace of tools/graph_probe.py, profiles/r06_graph_probe.txt).
            static const bool main_first = [] { const char* v = getenv("MF_JACOBI_MAIN_FIRST"); return v && *v == '1'; }();          // tuning aid
            hipStream_t s_piece = main_first ? side : st, s_main = main_first ? st : side;
            int rc = tail_waves == 8 ? launch_wave<OMEGA, 2, 8>(b, x, taps, lam, inv_on, F, S, iters, s_piece, S - r, r)
                                     : launch_wave<OMEGA, 3, 4>(b, x, taps, lam, inv_on, F, S, iters, s_piece, S - r, r);
            if (rc == MF_OK) rc = launch_wave<OMEGA, K, 1>(b, x, taps, lam, inv_on, F, S, iters, s_main, 0, S - r);
            const int rj = jacobi_side_join(st);
            return rc != MF_OK ? rc : rj;
        }
    }
    if (F <= 64 * K) MF_JACOBI(OMEGA, K, 1);
    if (F <= 128 * K) MF_JACOBI(OMEGA, K, 2);
    if (F <= 256 * K) MF_JACOBI(OMEGA, K, 4);
    if (F <= 512 * K) MF_JACOBI(OMEGA, K, 8);
    if (OMEGA == 10) {
        if (F <= 512 * 10) MF_JACOBI(10, 10, 8);
        if (F <= 512 * 19) MF_JACOBI(10, 19, 8);
    }
    if (OMEGA == 30 && F <= 512 * 19) MF_JACOBI(30, 19, 8);
#undef MF_JACOBI
    return MF_JACOBI_NOT_HERE;
}

}  // namespace

int MF_CAT(launch_jacobi_spec_g, MF_JACOBI_GROUP)(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S,
                                                  int omega, int iters, int want, hipStream_t st)
{
    constexpr int O0 = 8 * MF_JACOBI_GROUP;
    switch (omega - O0) {
#define MF_CASE(i) case i: return launch_radius<O0 + i>(b, x, taps, lam, inv_on, F, S, iters, want, st);
    MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6) MF_CASE(7) MF_CASE(8)
#undef MF_CASE
    default: return MF_JACOBI_NOT_HERE;
    }
}

// The tiled form (clips beyond 64 x 8 x 19 frames, jacobi_kernels.h) of the radii the BASELINE configs use; every other radius takes
// jacobi.hip's run-time-radius tiled kernel.  Defined once: in the group that owns radius 10.
#if MF_JACOBI_GROUP == 1
// (omega = 30: the 19-frames-per-lane loop spills under its 256-register budget (16-18 TFLOP/s at 20,000 frames); MF_JACOBI_TILE30 = 10
// takes the 10-frames-per-lane loop on tiles of 5,120 frames instead -- tuning aid, read once)
static int tile30_k()
{
    static const int k = [] { const char* v = getenv("MF_JACOBI_TILE30"); return v && atoi(v) == 19 ? 19 : 10; }();
    return k;
}
int jacobi_tiled_spec_len(int omega)
{
    if (omega == 10) return 64 * MF_JACOBI_TILE_WAVES * MF_JACOBI_TILE_K;
    if (omega == 30) return 64 * MF_JACOBI_TILE_WAVES * tile30_k();
    return 0;
}
int launch_jacobi_tiled_spec(const double* b, double* x, const double* taps, const double* lam, const double* inv_on, int F, int S, int omega,
                             int iters, const JacobiTile& tile, int ntiles, hipStream_t st)
{
    if (omega == 10) return launch_wave_tiled<10, MF_JACOBI_TILE_K, MF_JACOBI_TILE_WAVES>(b, x, taps, lam, inv_on, F, S, iters, tile, ntiles, st);
    if (omega == 30 && tile30_k() == 19) return launch_wave_tiled<30, 19, MF_JACOBI_TILE_WAVES>(b, x, taps, lam, inv_on, F, S, iters, tile, ntiles, st);
    if (omega == 30) return launch_wave_tiled<30, 10, MF_JACOBI_TILE_WAVES>(b, x, taps, lam, inv_on, F, S, iters, tile, ntiles, st);
    return MF_JACOBI_NOT_HERE;
}
#endif

}  // namespace mf

// Kernel 2b: per-frame, per-cell mesh warp with the crop-boundary scan folded in.
//
// Reference (meshflowstabilizer.py, "mfs.py"): for every frame, every mesh cell in row-major order warps a
// full-frame float64 mask (cv2.warpPerspective, mfs.py:1050-1052), transforms ALL pixel coordinates with
// the cell's inverse homography (cv2.perspectiveTransform, mfs.py:1054) and merges them into the frame's
// coordinate map under that mask (np.where, mfs.py:1060-1061); one cv2.remap (bilinear, constant border,
// mfs.py:1063-1069) and four edge scans (mfs.py:1075-1098) follow.  O(R*C*H*W) per frame.
//
// This kernel computes the same result in one pass over output pixels:
//   owner(pixel) = LAST cell in row-major order whose mask test passes  (painter merge)
//   (u, v)       = float32( Hi_owner * (x, y, 1) ), or (W+1, H+1) when no cell covers the pixel
//   crop flags   = |u| < 1, |u-(W-1)| < 1, |v| < 1, |v-(H-1)| < 1  -> atomic max/min per frame
//   out          = cv2.remap fixed-point bilinear of the source frame at (u, v)
// Coordinates are float64 exactly as OpenCV evaluates them (no FMA contraction: -ffp-contract=off), then
// float32, then 1/32-pixel fixed point; interpolation is integer.  The result is bit-identical to the CPU
// oracle (oracle/warp_oracle.c).
//
// Mapping (gfx950).  One wavefront = one workgroup = one 32 x 8 pixel "footprint" of one frame; a lane owns 4 consecutive pixels
// of one row (12 contiguous output bytes -> one global_store_dwordx3).  Grid = (8 * per_xcd, frames): footprints are handed out
// XCD-aware (mf_common.h WarpGeom).  Per footprint:
//   1. footprint_plan_kernel (cell_table.hip) wrote the wave-uniform PLAN (two scalar loads): up to 8 candidate cells in
//      descending order, each IN (all 256 pixels pass its mask test) or MIXED, with -- for short lists -- the one or two mask
//      edges that can fail; the footprint's SOURCE REGION, the window of the source frame that holds every bilinear tap of
//      every pixel, ready-made as LDS origin + first dword in the frame; and certificates: STAGED, DEEP (the footprint lies in
//      the frame, every pixel has an owner and every tap is interior), UNIT (the projective denominator stays in (0.52, 1.9)
//      and varies slowly enough for the reciprocal guess), and the two path bits HOT and PAIR.
//   2. The window goes to LDS asynchronously: two global_load_lds_dwordx4 per lane, issued first, awaited after the
//      coordinate arithmetic.
//   3. HOT (one IN cell, everything certified; ~65 % of footprints at config-2 geometry): straight-line code.  The cell's Hi
//      comes in through scalar loads; 1/w: pixel 0 by v_rcp_f64 + Newton + the residual correction that makes it the IEEE
//      quotient; pixels 1..3 start from pixel 0's reciprocal (second-order guess + ONE Newton step + the same correction).
//   4. PAIR (two cells, ~27 %): the later cell wins where ONE of its mask edges passes (one float32 fma per pixel), the other
//      owns the rest; both Hi go to LDS by global->LDS DMA (one 80-byte load per cell, scalar base address) and a pixel's
//      owner is the byte offset of its matrix row.  A pixel inside the float32 error band of the edge sends the wavefront to 5.
//      MULTI (two to four cells with one- or two-edge codes -- the four cells around a mesh vertex; ~3 %, 14 % with a 32 x 32
//      mesh): the same with up to two edge functions per cell; coverage is checked at run time (a pixel without owner -> 5).
//   5. Everything else: the general last-cell-first loop over the list.  A pixel inside the error band of an edge is decided
//      by a division-free float64 comparison, and by OpenCV's exact arithmetic (division, rint) only within 1e-6 of the edge;
//      "no owner" is a ninth LDS row holding the matrix that maps every pixel to (W+1, H+1) -- the coordinate code has no
//      special case.  Footprints the plan could not certify (frame border, uncovered pixels, oversized or unaligned windows)
//      check per pixel and use either two unaligned 8-byte global loads per pixel or the per-tap path with border colour and
//      crop flags (branch-free: loads at positions clamped into the frame, border colour selected afterwards).
//   6. cv2.remap: sx = rint(32u) via one fma against 1.5*2^23; taps = LDS byte loads straight into the blend's layout (the two
//      horizontal neighbours of a channel in the 16-bit halves of a register); v_mul_u32_u24 + v_mad_u32_u24 lerp both halves
//      vertically at once, v_dot2_u32_u16 lerps horizontally with weights scaled so that the rounded byte lands in byte 2; six
//      v_perm_b32 + three v_or_b32 gather the lane's 12 output bytes.
// Algorithmic HBM traffic: 2*H*W*3 bytes per frame (each source byte read once, each output byte written once); measured
// 1.05x that.  No dense contraction: no MFMA.  What bounds it: vector AND scalar instruction issue, not HBM -- DESIGN.md
// section 4.3, profiles/r02_phase_profile.txt.  Everything a wavefront needs before its pixels is therefore host-made
// (WarpGeom) or plan-made (FootRegion): the hot path issues 222 vector and 74 scalar instructions per wavefront.  The float32
// edge functions come scaled by their own evaluation error bound (cell_table.hip): beyond +-1 their sign is exact.
#include "mf_common.h"
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <type_traits>

// At most 80 scalar registers: a CU admits min(8, 800 / (ceil(sgpr / 16) * 16 + 16)) workgroups of 256 threads
// (MI355X_MICROARCH.md), i.e. 7 with the 94 the compiler would take and 8 with 80 (the excess is kept in VGPR lanes, the kernel
// stays at 64 VGPRs): -1.7 % kernel time.
#define MF_WARP_ATTR __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))

namespace mf {

// The cell table is written by earlier kernels and only read here: pointers into it live in the constant address space, so
// that wave-uniform reads stay scalar loads (s_load) whatever else the kernel does (the global->LDS copies count as memory
// writes for the compiler, which otherwise turns later record reads into per-lane vector loads and spends 40 VGPRs on them).
typedef const __attribute__((address_space(4))) double* crec_t;
typedef const __attribute__((address_space(4))) float* cedge_t;

// Workgroup = ONE wavefront (its tile = its 32 x 8 footprint).  Wavefronts never cooperate (no barrier, no shared LDS data), and
// a multi-wave workgroup keeps the slots of its finished wavefronts until the slowest one -- often on a slower ownership
// path -- is done: 4 x 1 wavefronts 1.516 ms (cfg2) / 3.410 (cfg3), 2 x 1: 1.505 / 3.392, 1 x 1: 1.492 / 3.326; 4 x 2 and 4 x 4
// (fewer dispatches) 1.65 / 1.89.
// (Round 5, at the final kernels -- the launch rate is per WORKGROUP, an empty kernel of 4-wave workgroups launches 4 x as many
// wavefronts per ns, tools/ubench_launch.hip -- 2 / 4 wavefronts per workgroup again: config 2 +3.6 / +4.1 %, config 3 +4.7 / +4.3 %,
// 4K +3.4 / +4.1 %, an all-hot footprint stream +-0: the dispatcher is not what the kernel waits for.)
// More than one footprint per wavefront (a vertical stack, or a run along x with the next footprint's plan and window prefetched
// into a second LDS buffer behind counted vmcnt waits) is slower as well: 2 per wavefront +4 %, 4 per wavefront +9 %.
// (Round 5: a wavefront that takes the hot footprint BELOW its own as well when both have the same owner -- one plan round trip, one
// matrix, two windows, two batches of pixels, everything else through the regular code one footprint after the other; zero scratch,
// byte-identical -- all-hot stream -0.7 %, 4K -0.2 %, config 2 +2.4 %, config 3 +5.2 %: what a wavefront does once per footprint is
// not what bounds the kernel.  DESIGN.md section 4.3.)
constexpr int FOOT_W = MF_FOOT_W;   // 8 lanes x 4 pixels
constexpr int FOOT_H = MF_FOOT_H;   // 64 lanes / 8
constexpr int MAX_MESH = 64;    // R, C <= 64
// The float32 edge functions are stored scaled by their own evaluation error bound (cell_table.hip): beyond +-1 their sign is the
// exact function's sign; inside the band the float64 comparison decides.
constexpr float EDGE_BAND = 1.0f;
// A pixel's owner is kept as the byte offset of the owner's row in the wavefront's s_hi block (80-byte rows, one per list
// entry).  Row 8 holds the matrix {0, 0, W+1; 0, 0, H+1; 0, 0, 1}: a pixel no cell covers runs through the same arithmetic and
// comes out at exactly (W+1, H+1) (mfs.py:983-984) -- no special case, no select, in the coordinate code.
constexpr uint32_t OWN_ROW = 80, OWN_NONE = 8 * OWN_ROW;
constexpr int LDS_PITCH = MF_STAGE_PITCH;
constexpr int LDS_WINDOW_BYTES = MF_STAGE_CHUNKS * 16;
// In front of the window: room for the LDS row of frame row -1 (and the pixel of column -1 in front of it) that the border path paints
// in the border colour; the row of frame row H lands behind row 11, inside the window's own bytes.
constexpr int LDS_WINDOW_PAD = 176;

// LDS pointer of a __shared__ object WITHOUT the generic -> LDS conversion (which comes with a null check: s_mov src_shared_base + s_cmp +
// s_cselect, three scalar instructions per global->LDS copy, and the scalar unit is as loaded as the vector unit here): the low
// 32 bits of a generic address into LDS are the LDS address.
typedef __attribute__((address_space(3))) uint8_t* lds_bytes_t;
__device__ __forceinline__ lds_bytes_t lds_ptr(const void* shared_object)
{
    return (lds_bytes_t)(uintptr_t)(uint32_t)(uintptr_t)shared_object;
}

// a * b + c on the 24-bit multiplier.  The empty asm makes `c` opaque so that the compiler keeps two chained
// v_mad_u32_u24 instead of re-associating them into mul + mul + add3 (no instruction is emitted by it, so the
// compiler still pads every hazard itself).
__device__ __forceinline__ uint32_t umad24(uint32_t a, uint32_t b, uint32_t c)
{
    asm("" : "+v"(c));
    return __umul24(a, b) + c;
}

// min(a, b, c) in ONE instruction (the compiler re-associates a chain of min() into more v_min_u32 than needed)
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// bits [5..28] of the raw float 32u + 1.5*2^23 are (sx >> 5) + MAGIC_HI for 0 <= sx < 2^22
constexpr uint32_t MAGIC_HI = (0x4B400000u >> 5) & 0xFFFFFFu;

// a.lo * b.lo + a.hi * b.hi + c on 16-bit halves (v_dot2_u32_u16)
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

__device__ __forceinline__ int cv_round_f32(float v)
{
    const float r = rintf(v);
    return (r >= -2147483648.0f && r < 2147483648.0f) ? (int)r : (int)0x80000000;
}

// OpenCV's mask test, exactly (imgwarp.cpp WarpPerspectiveInvoker: 64-wide destination blocks).
// (OpenCV's block is min(1024 / min(16, H), W) pixels wide: 64 for every frame of 16 rows or more -- or narrower than 64 pixels, which
// is one block either way.  A frame under 16 rows tall AND over 64 pixels wide would get wider blocks, i.e. one rounding of x-dependent
// terms placed differently: visible only on an exact rounding tie at a mask edge.  Not modelled -- here, in oracle/warp_oracle.c and in
// oracle/meshflow_oracle.py alike; tests/test_cv2_crosscheck.py is where a real OpenCV would show it.)
__device__ __forceinline__ bool mask_test_exact(const double* __restrict__ M, int lo_x, int hi_x, int lo_y, int hi_y,
                                             int x, int y)
{
    const double xb = (double)(x & ~63), x1 = (double)(x & 63), yy = (double)y;
    const double X0 = (M[0] * xb + M[1] * yy) + M[2];
    const double Y0 = (M[3] * xb + M[4] * yy) + M[5];
    const double W0 = (M[6] * xb + M[7] * yy) + M[8];
    const double Wd = W0 + M[6] * x1;
    const double Ws = Wd != 0.0 ? 32.0 / Wd : 0.0;
    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + M[0] * x1) * Ws));
    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + M[3] * x1) * Ws));
    const int X = (int)rint(fX);
    const int Y = (int)rint(fY);
    // non-zero bilinear sample of the 255-filled rect <=> a tap with non-zero weight lies on it
    return X > lo_x && X < hi_x && Y > lo_y && Y < hi_y;
}

// 1/w with the exact bits of IEEE division for 0.5 <= |w| <= 2: the compiler's own f64 division sequence
// (v_div_scale / v_rcp / 2 Newton steps / residual / v_div_fmas / v_div_fixup) without the scaling and
// special-case steps, which are the identity in that range.  tests/test_gpu_parity.py checks it against
// 1.0 / w on random inputs (mf_selftest_recip).
__device__ __forceinline__ double recip_unit_range(double w)
{
    double r = __builtin_amdgcn_rcp(w);
    double e = __builtin_fma(-w, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-w, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-w, r, 1.0);
    return __builtin_fma(e, r, r);
}

// 1/w for the lane's pixels 1..3 WITHOUT v_rcp_f64 (16 issue cycles) and with one Newton step less: the denominators of
// consecutive pixels differ by h6 (w_j = w_0 + j h6 up to rounding), so with r0 = 1/w_0
//     1/w_j = r0 (1 - e + e^2 - ...),  e = j h6 r0,
// and the second-order guess g = r0 - j c1 + j^2 c2 (c1 = h6 r0^2, c2 = h6^2 r0^3) is within e^3 (1 + e) of 1/w_j.  One Newton
// step squares that; the residual-correction step of recip_unit_range then delivers the correctly rounded quotient exactly as
// it does there, where its input is also an approximation good to about one ulp.  The caller guarantees |c1| <= 2.5e-4, i.e.
// e <= 3 |c1| / |r0| <= 1.5e-3 (|r0| > 1/2), so the Newton step leaves a relative error below (1.002 * 3.4e-9)^2 < 2^-56.
// mf_selftest_recip checks it against IEEE division on hashed (w_0, h6, j).
constexpr double RECIP_GUESS_LIMIT = 2.5e-4;
__device__ __forceinline__ double recip_guess(double r0, double c1, double c2, double j)
{
    return __builtin_fma(j * j, c2, __builtin_fma(-j, c1, r0));
}
__device__ __forceinline__ double recip_from_guess(double w, double g)
{
    double e = __builtin_fma(-w, g, 1.0);
    g = __builtin_fma(g, e, g);
    e = __builtin_fma(-w, g, 1.0);
    return __builtin_fma(e, g, g);
}

// Source coordinates of the lane's four pixels under cell `rec`'s inverse homography:
// cv2.perspectiveTransform (matmul.simd.hpp) -- float32 point, float64 matrix, float32 result.
// SELECT = false: every pixel takes the new coordinates; true: only those in `pass`.
// `certified` (wave-uniform): the plan has checked on the footprint's corners that the denominator stays inside (0.52, 1.9) and
// that the reciprocal guess applies (MF_PLAN_UNIT) -- both tests are then skipped.
template <bool SELECT>
__device__ __forceinline__ void cell_coords(crec_t rec, double xs0, double yy, int x0, uint32_t pass,
                                            float (&u)[4], float (&v)[4], bool certified = false)
{
    (void)x0;
    double Hi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hi[i] = rec[MF_CELL_OFF_HI + i];
    const double t6 = yy * Hi[7], t0 = yy * Hi[1], t3 = yy * Hi[4];
    double w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = ((xs0 + (double)j) * Hi[6] + t6) + Hi[8];     // (xs0 + j is exact: small integers)
    // pixel 0: full reciprocal; pixels 1..3 start from it (recip_guess).  A cell whose denominator leaves [0.5, 2) or
    // changes too fast along x for the guess (strong perspective: |h6| / w^2 > 2.5e-4 per pixel) takes the generic division.
    bool fast_ok = certified;
    if (!certified) {
        uint32_t eor = 0;                                      // |w| in [0.5, 2) <=> frexp exponent in {0, 1}
#pragma unroll
        for (int j = 0; j < 4; ++j) eor |= (uint32_t)__builtin_amdgcn_frexp_exp(w4[j]);
        // (the test |h6| <= limit * w0^2 is the same condition as |c1| <= limit without waiting for the reciprocal)
        const bool guess_ok = fabs(Hi[6]) <= (0.96 * RECIP_GUESS_LIMIT) * (w4[0] * w4[0]);
        fast_ok = __ballot(eor > 1u || !guess_ok) == 0;
    }
    if (fast_ok) {
        const double iw0 = recip_unit_range(w4[0]);
        const double c1 = Hi[6] * (iw0 * iw0), c2 = (Hi[6] * c1) * iw0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double xs = xs0 + (double)j;
            const double iw = j == 0 ? iw0 : recip_from_guess(w4[j], recip_guess(iw0, c1, c2, (double)j));
            const float un = (float)(((xs * Hi[0] + t0) + Hi[2]) * iw);
            const float vn = (float)(((xs * Hi[3] + t3) + Hi[5]) * iw);
            if (SELECT) {
                const bool p = (pass >> j) & 1u;
                u[j] = p ? un : u[j];
                v[j] = p ? vn : v[j];
            } else {
                u[j] = un;
                v[j] = vn;
            }
        }
    } else {                                                   // far-from-affine cell: generic division
        // (unrolled: a rolled loop indexes u[] / v[] by select chains, and their initial values -- eight moves -- are then
        // hoisted in front of the branch, onto the fast path)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double xs = xs0 + (double)j;
            const double w = w4[j];
            const bool ok = fabs(w) > 1.1920928955078125e-07;
            const double iw = 1.0 / w;
            const float un = ok ? (float)(((xs * Hi[0] + t0) + Hi[2]) * iw) : 0.0f;
            const float vn = ok ? (float)(((xs * Hi[3] + t3) + Hi[5]) * iw) : 0.0f;
            const bool p = !SELECT || ((pass >> j) & 1u);
            u[j] = p ? un : u[j];
            v[j] = p ? vn : v[j];
        }
    }
}

// (A/B switches of tools/ab_warp.py: the lane-uniform pair path and the multi path's cheap chain are built on the FAST64 helpers)
#if defined(MF_NO_FAST64) && !defined(MF_NO_FASTPAIR)
#define MF_NO_FASTPAIR 1
#endif
#if defined(MF_NO_FAST64) && !defined(MF_NO_FASTMULTI)
#define MF_NO_FASTMULTI 1
#endif
#ifndef MF_NO_FAST64
// FAST COORDINATES.  cv2.perspectiveTransform's float64 chain -- (x h0 + y h1) + h2 with every product and sum rounded, the
// correctly rounded 1 / w, the rounded product -- only matters through its float32 conversion.  A cheaper float64 chain (fused
// affine forms, ONE reciprocal of the lane's four denominators refined by ONE Newton step) lands within 118 float64 ulps of the exact
// chain's value (bound: DESIGN.md section 4.3, certified per footprint by the plan: MF_PLAN_FAST64; mf_selftest_fast64_margin measures
// the distance), so both convert to the SAME float32 unless the cheap value lies within that distance of a float32 rounding midpoint,
// i.e. unless the low 29 mantissa bits are within FAST64_WINDOW (4.3 x the bound) of 0x10000000.
// midpoint_key() is below FAST64_NEAR exactly then (one v_lshl_add_u32 on the low dword); a wavefront with any such value redoes its
// coordinates with the exact chain (about one wavefront in 1,000 at config-2 geometry).
constexpr uint32_t FAST64_WINDOW = 512u;
// (low dword << 3) + const: the 29 dropped mantissa bits, shifted to the top of the register and offset so that the window around the
// midpoint pattern 0x10000000 maps to [0, 16 FAST64_WINDOW) -- ONE v_lshl_add_u32 per value; the smallest key of a lane decides.
constexpr uint32_t FAST64_NEAR = 16u * FAST64_WINDOW;
__device__ __forceinline__ uint32_t midpoint_key(double a)
{
    return ((uint32_t)__double_as_longlong(a) << 3) + ((0x10000000u + FAST64_WINDOW) << 3);
}

// Quotients n_j / w_j and m_j / w_j of a lane's four pixels on the cheap chain, whatever matrices the forms came from: ONE reciprocal
// for the four denominators -- R = 1 / (w0 w1 w2 w3) by v_rcp_f64 + ONE Newton step (0.07 < product < 13.1), then 1 / w0 = (R w2 w3) w1
// and so on: nine multiplications; the rounding errors of the w_j themselves cancel (the same values sit in the product), what remains
// is 5 roundings per reciprocal plus what the Newton step leaves: v_rcp_f64 is good to 2^-24.36 (tools/ubench_semantics.hip: 2^26
// evenly spaced mantissas x 8 exponents, profiles/r06_ubench_semantics.txt), one step squares that: 2^-48.7 = 20 u (u = 2^-53) -- a
// second step (rounds 5-6a) took it to 1 u for two more float64 instructions per lane.  Returns the smallest midpoint key of the eight values.
__device__ __forceinline__ uint32_t cheap_quotients(const double (&w)[4], const double (&n)[4], const double (&m)[4], float (&u)[4], float (&v)[4],
                                                    uint32_t* keys = nullptr, double* raw = nullptr)
{
    const double q01 = w[0] * w[1], q23 = w[2] * w[3], pr = q01 * q23;
    double r = __builtin_amdgcn_rcp(pr);
    double e = __builtin_fma(-pr, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double ra = r * q23, rb = r * q01;
    const double g[4] = { ra * w[1], ra * w[0], rb * w[3], rb * w[2] };
    uint32_t key = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double a = n[j] * g[j], b = m[j] * g[j];
        u[j] = (float)a;
        v[j] = (float)b;
        key = j == 0 ? min(midpoint_key(a), midpoint_key(b)) : umin3(key, midpoint_key(a), midpoint_key(b));
        if (keys) { keys[2 * j] = midpoint_key(a); keys[2 * j + 1] = midpoint_key(b); }
        if (raw) { raw[2 * j] = a; raw[2 * j + 1] = b; }
    }
    return key;
}

// The cheap chain for a lane whose four pixels step along x (VERT = false: (x0 + j, y0)) or along y (VERT: (x0, y0 + j), the
// transposed lane mapping of the pair path); returns the smallest midpoint key (< FAST64_NEAR = some value too close to a float32 midpoint).
// `keys` (self-test only): the eight midpoint keys, u then v per pixel.
template <bool VERT>
__device__ __forceinline__ uint32_t coords_fast_dir(const double (&Hi)[9], double xs0, double yy0, float (&u)[4], float (&v)[4], uint32_t* keys = nullptr,
                                                    double* raw = nullptr)
{
    const double t0 = VERT ? yy0 : xs0, o = VERT ? xs0 : yy0;                        // stepping coordinate, the other one
    const double a0 = Hi[VERT ? 1 : 0], a3 = Hi[VERT ? 4 : 3], a6 = Hi[VERT ? 7 : 6];   // coefficients of the stepping coordinate
    const double c0 = __builtin_fma(o, Hi[VERT ? 0 : 1], Hi[2]), c3 = __builtin_fma(o, Hi[VERT ? 3 : 4], Hi[5]), c6 = __builtin_fma(o, Hi[VERT ? 6 : 7], Hi[8]);
    // the affine forms at the lane's first pixel, then + j a (j = 1, 2, 3 are exact constants): one fma per pixel and form
    double w[4], n[4], m[4];
    w[0] = __builtin_fma(t0, a6, c6); n[0] = __builtin_fma(t0, a0, c0); m[0] = __builtin_fma(t0, a3, c3);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        w[j] = __builtin_fma((double)j, a6, w[0]);
        n[j] = __builtin_fma((double)j, a0, n[0]);
        m[j] = __builtin_fma((double)j, a3, m[0]);
    }
    return cheap_quotients(w, n, m, u, v, keys, raw);
}
// The hot path's coordinates by the cheap chain; false (wave-uniform) when some value is too close to a float32 midpoint.
__device__ __forceinline__ bool coords_fast(const double (&Hi)[9], double xs0, double yy, float (&u)[4], float (&v)[4], uint32_t* keys = nullptr)
{
    return __ballot(coords_fast_dir<false>(Hi, xs0, yy, u, v, keys) < FAST64_NEAR) == 0;
}
__device__ __forceinline__ bool cell_coords_fast(crec_t rec, double xs0, double yy, float (&u)[4], float (&v)[4])
{
    double Hi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hi[i] = rec[MF_CELL_OFF_HI + i];
    return coords_fast(Hi, xs0, yy, u, v);
}
#endif

// Per-pixel mask test of a MIXED cell for the lane's four pixels; returns the 4-bit pass mask.
// Division-free decision: with Xn = M0 x + M1 y + M2 and Wd = M6 x + M7 y + M8 > 0, OpenCV's
// fX = fl(Xn * fl(32/Wd)) differs from 32 Xn / Wd by < 1e-9 relative, and rint(fX) > lo <=> fX > lo + 1/2
// (lo is even).  So the sign of q = 32 Xn - (lo + 1/2) Wd (and its three siblings) decides the test unless
// |q| <= 1e-6 Wd; only then is the exact arithmetic (division, rint) needed.
__device__ __forceinline__ uint32_t cell_mask_test(crec_t rec, double xs0, double yy, int x0, int y,
                                                   uint32_t unowned)
{
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = rec[MF_CELL_OFF_M + i];
    const double rL = rec[MF_CELL_OFF_RECT + 0], rT = rec[MF_CELL_OFF_RECT + 1];
    const double rR = rec[MF_CELL_OFF_RECT + 2], rB = rec[MF_CELL_OFF_RECT + 3];
    const double loxh = 32.0 * (rL - 1.0) + 0.5, hixh = 32.0 * (rR + 1.0) - 0.5;
    const double loyh = 32.0 * (rT - 1.0) + 0.5, hiyh = 32.0 * (rB + 1.0) - 0.5;
    const double RX = __builtin_fma(M[1], yy, M[2]);
    const double RY = __builtin_fma(M[4], yy, M[5]);
    const double RW = __builtin_fma(M[7], yy, M[8]);
    uint32_t ok = 0, amb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double xs = xs0 + (double)j;
        const double Wd = __builtin_fma(M[6], xs, RW);
        const double X32 = 32.0 * __builtin_fma(M[0], xs, RX);
        const double Y32 = 32.0 * __builtin_fma(M[3], xs, RY);
        const double qmin = fmin(fmin(__builtin_fma(-loxh, Wd, X32), __builtin_fma(hixh, Wd, -X32)),
                                 fmin(__builtin_fma(-loyh, Wd, Y32), __builtin_fma(hiyh, Wd, -Y32)));
        const double t = 1e-6 * Wd;
        const bool sane = (Wd > 0.25) & (Wd < 4.0);
        const bool yes = sane & (qmin > t), no = sane & (qmin < -t);
        ok |= yes ? (1u << j) : 0u;
        amb |= (yes | no) ? 0u : (1u << j);
    }
    amb &= unowned;
    if (__ballot(amb != 0) != 0) {                             // rare: a pixel within 1e-6 of a mask edge
        const int lo_x = 32 * ((int)rL - 1), hi_x = 32 * ((int)rR + 1);
        const int lo_y = 32 * ((int)rT - 1), hi_y = 32 * ((int)rB + 1);
#pragma unroll 1
        for (int j = 0; j < 4; ++j)
            if (((amb >> j) & 1u) && mask_test_exact(M, lo_x, hi_x, lo_y, hi_y, x0 + j, y)) ok |= 1u << j;
    }
    return ok & unowned;
}

// cv2.remap's fixed point: sx = rint(32 u) by the 1.5*2^23 trick -- the fma rounds 32u + magic once, to nearest even, and the integer
// sits in the low mantissa bits (valid for |32u| < 2^22; anything else lands far outside the "deep interior" window and is redone
// exactly by the generic path).  Raw float bits of 32u + 1.5*2^23: the low 22 bits hold sx for 0 <= sx < 2^22.
__device__ __forceinline__ void fixed_point(const float (&u)[4], const float (&v)[4], uint32_t (&bx)[4], uint32_t (&by)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bx[j] = __float_as_uint(__builtin_fmaf(u[j], 32.0f, 12582912.0f));
        by[j] = __float_as_uint(__builtin_fmaf(v[j], 32.0f, 12582912.0f));
    }
}

// Taps + blend of a footprint with a staged window.  The taps come from the staged window by BYTE loads with immediate offsets, already in the
// layout the blend wants -- per pixel and channel the two horizontal neighbours in the 16-bit halves of one register (X0 | X1 << 16),
// for rows iy and iy + 1: ds_read_u8 delivers X0 in the low byte of one register, ds_read_u8_d16_hi X1 in bits 16-23 of another
// (with SRAM ECC a d16 load zeroes the other half instead of preserving it: measured), and one v_or_b32 (a 2-cycle instruction) joins
// them.  Against three ds_read2_b32 + four v_alignbyte_b32 + six v_perm_b32 per pixel that is 28 VALU issue cycles per pixel less
// (the LDS pipe takes 12 byte loads per pixel instead; it has the room).  The compiler does not see these loads, so the waits are
// placed here.
struct TapRegs { uint32_t lo[6], hi[6]; };      // {B, G, R} of row iy, then of row iy + 1: X0 in lo (byte 0), X1 in hi (byte 2)

// Tap address of one pixel: LDS byte address of its top-left tap's B.
template <int PITCH>
__device__ __forceinline__ uint32_t tap_address(uint32_t bxj, uint32_t byj, uint32_t lds_origin)
{
    // ix = bits[5..21] of the raw float; bits[22..28] (the 1.5*2^23 pattern, constant) ride along in the 24-bit multiplier
    // operand and are taken out again through the origin
    return umad24(byj >> 5, (uint32_t)PITCH, umad24(bxj >> 5, 3u, 0u - lds_origin - MAGIC_HI * (3u + (uint32_t)PITCH)));
}

// The 24 byte loads of TWO pixels and their wait in ONE asm block: the compiler does not see LDS loads issued from inline asm, so nothing
// -- no copy, no spill, no reordering under another compiler version or flag -- can come between a load and the wait that makes its
// register valid.  (Row pitch in the immediates: one instantiation per window layout.)
#define MF_TAP_LOADS(R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11, A, P0, P1, P2, P3, P4, P5)                                    \
    "ds_read_u8 " R0 ", " A " offset:0\n\tds_read_u8_d16_hi " R1 ", " A " offset:3\n\t"                                              \
    "ds_read_u8 " R2 ", " A " offset:1\n\tds_read_u8_d16_hi " R3 ", " A " offset:4\n\t"                                              \
    "ds_read_u8 " R4 ", " A " offset:2\n\tds_read_u8_d16_hi " R5 ", " A " offset:5\n\t"                                              \
    "ds_read_u8 " R6 ", " A " offset:" P0 "\n\tds_read_u8_d16_hi " R7 ", " A " offset:" P3 "\n\t"                                    \
    "ds_read_u8 " R8 ", " A " offset:" P1 "\n\tds_read_u8_d16_hi " R9 ", " A " offset:" P4 "\n\t"                                    \
    "ds_read_u8 " R10 ", " A " offset:" P2 "\n\tds_read_u8_d16_hi " R11 ", " A " offset:" P5 "\n\t"
#define MF_TAP_PAIR_ASM(P0, P1, P2, P3, P4, P5)                                                                                     \
    asm volatile(MF_TAP_LOADS("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7", "%8", "%9", "%10", "%11", "%24", P0, P1, P2, P3, P4, P5)   \
                 MF_TAP_LOADS("%12", "%13", "%14", "%15", "%16", "%17", "%18", "%19", "%20", "%21", "%22", "%23", "%25", P0, P1, P2, P3, P4, P5) \
                 "s_waitcnt lgkmcnt(0)"                                                                                             \
                 : "=&v"(t.lo[0]), "=&v"(t.hi[0]), "=&v"(t.lo[1]), "=&v"(t.hi[1]), "=&v"(t.lo[2]), "=&v"(t.hi[2]),                  \
                   "=&v"(t.lo[3]), "=&v"(t.hi[3]), "=&v"(t.lo[4]), "=&v"(t.hi[4]), "=&v"(t.lo[5]), "=&v"(t.hi[5]),                  \
                   "=&v"(u.lo[0]), "=&v"(u.hi[0]), "=&v"(u.lo[1]), "=&v"(u.hi[1]), "=&v"(u.lo[2]), "=&v"(u.hi[2]),                  \
                   "=&v"(u.lo[3]), "=&v"(u.hi[3]), "=&v"(u.lo[4]), "=&v"(u.hi[4]), "=&v"(u.lo[5]), "=&v"(u.hi[5])                   \
                 : "v"(at0), "v"(at1) : "memory")
template <int PITCH>
__device__ __forceinline__ void taps_pair(uint32_t at0, uint32_t at1, TapRegs& t, TapRegs& u)
{
    static_assert(PITCH == MF_STAGE_PITCH || PITCH == MF_COMPACT_PITCH, "one asm string per window pitch");
    static_assert(MF_STAGE_PITCH == 160 && MF_COMPACT_PITCH == 112, "the immediate offsets below are the pitch + 0..5");
    if (PITCH == MF_STAGE_PITCH) MF_TAP_PAIR_ASM("160", "161", "162", "163", "164", "165");
    else MF_TAP_PAIR_ASM("112", "113", "114", "115", "116", "117");
}

// (Round 6, measured and dropped, profiles/r06_ab_trims.txt: the four weights as two packed pairs -- v_pk_mad_u16 with the clamp bit for
// 64 (32 - fx)(32 - fy) = 65536 -> 65535, v_pk_mul_lo_u16 -- and two chained v_dot2_u32_u16 per channel instead of v_mul + v_mad + dot2:
// 16 issue cycles per wavefront less by the table, byte-identical, +0.7...1.6 % SLOWER; and the tap address as two hand-placed
// v_mad_u32_u24: 8 cycles less, -0.3 % / -0.3 % / +1.6 %.  Neither the issue-cycle table nor the energy table (profiles/r03_ubench_power.txt) predicts that; cause not identified.)
__device__ __forceinline__ void blend_pixel(uint32_t bxj, uint32_t byj, const TapRegs& t, uint32_t& oB, uint32_t& oG, uint32_t& oR)
{
    // vertical lerp of both 16-bit fields at once (each <= 255 * 32: no carry between them)
    const uint32_t fy = byj & 31u, wy = 32u - fy;
    const uint32_t vB = umad24(t.lo[3] | t.hi[3], fy, __umul24(t.lo[0] | t.hi[0], wy));
    const uint32_t vG = umad24(t.lo[4] | t.hi[4], fy, __umul24(t.lo[1] | t.hi[1], wy));
    const uint32_t vR = umad24(t.lo[5] | t.hi[5], fy, __umul24(t.lo[2] | t.hi[2], wy));
    // horizontal lerp: v_dot2_u32_u16 with the weight pair (32 - fx, fx) scaled by 64, so that ((sum + 512) >> 10) lands in byte 2:
    // (sum + 512) * 64 < 2^24
    const uint32_t fx = bxj & 31u;
    const uint32_t wq = umad24(fx, 0x3FFFC0u, 2048u);           // 64 (32 - fx) | 64 fx << 16
    oB = udot2(vB, wq, 32768u);
    oG = udot2(vG, wq, 32768u);
    oR = udot2(vR, wq, 32768u);
}

// The 2 x 2 taps of ONE pixel from four separate LDS positions (the per-tap path of frame-border footprints: every tap at its position
// clamped into the frame, a00 / a01 = row iy at columns ix / ix + 1, a10 / a11 = row iy + 1), in the blend's layout.  Loads and wait in
// one asm block: nothing can be scheduled between them.
__device__ __forceinline__ void taps_clamped(uint32_t a00, uint32_t a01, uint32_t a10, uint32_t a11, TapRegs& t)
{
    asm volatile("ds_read_u8 %0, %12 offset:0\n\tds_read_u8_d16_hi %1, %13 offset:0\n\t"
                 "ds_read_u8 %2, %12 offset:1\n\tds_read_u8_d16_hi %3, %13 offset:1\n\t"
                 "ds_read_u8 %4, %12 offset:2\n\tds_read_u8_d16_hi %5, %13 offset:2\n\t"
                 "ds_read_u8 %6, %14 offset:0\n\tds_read_u8_d16_hi %7, %15 offset:0\n\t"
                 "ds_read_u8 %8, %14 offset:1\n\tds_read_u8_d16_hi %9, %15 offset:1\n\t"
                 "ds_read_u8 %10, %14 offset:2\n\tds_read_u8_d16_hi %11, %15 offset:2\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t.lo[0]), "=&v"(t.hi[0]), "=&v"(t.lo[1]), "=&v"(t.hi[1]), "=&v"(t.lo[2]), "=&v"(t.hi[2]),
                   "=&v"(t.lo[3]), "=&v"(t.hi[3]), "=&v"(t.lo[4]), "=&v"(t.hi[4]), "=&v"(t.lo[5]), "=&v"(t.hi[5])
                 : "v"(a00), "v"(a01), "v"(a10), "v"(a11) : "memory");
}

// (two pixels' loads in flight at a time: 24 registers; a software pipeline with counted lgkmcnt waits measured the same.  Round 6: the
// taps as 16-bit loads -- three per tap row instead of six byte loads, one v_perm_b32 per channel and row instead of a v_or_b32 -- are
// byte-identical and 3.9 x slower: a ds_read_u16 at an ODD byte address costs 56 cycles per wave64 instruction against 1.9 at an even
// one, and a tap row starts at byte 3 ix; tools/ubench_lds_u16.hip, profiles/r06_ubench_lds_u16.txt.)
template <int PITCH = LDS_PITCH>
__device__ __forceinline__ void gather_blend_sums(const uint32_t (&bx)[4], const uint32_t (&by)[4], uint32_t lds_origin,
                                                  uint32_t (&oB)[4], uint32_t (&oG)[4], uint32_t (&oR)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        TapRegs t0, t1;
        taps_pair<PITCH>(tap_address<PITCH>(bx[j], by[j], lds_origin), tap_address<PITCH>(bx[j + 1], by[j + 1], lds_origin), t0, t1);
        blend_pixel(bx[j], by[j], t0, oB[j], oG[j], oR[j]);
        blend_pixel(bx[j + 1], by[j + 1], t1, oB[j + 1], oG[j + 1], oR[j + 1]);
    }
}
template <int PITCH = LDS_PITCH>
__device__ __forceinline__ uint3 gather_blend_staged(const uint32_t (&bx)[4], const uint32_t (&by)[4], uint32_t lds_origin)
{
    uint32_t oB[4], oG[4], oR[4];
    gather_blend_sums<PITCH>(bx, by, lds_origin, oB, oG, oR);
    // the 12 result bytes sit in byte 2 of the 12 sums: 6 v_perm_b32 + 3 v_or_b32 gather them into B0 G0 R0 B1 | G1 R1 B2 G2 |
    // R2 B3 G3 R3
    const uint32_t pair = 0x0C0C0602u, pair_hi = 0x06020C0Cu;
    uint3 d;
    d.x = __builtin_amdgcn_perm(oB[1], oR[0], pair_hi) | __builtin_amdgcn_perm(oG[0], oB[0], pair);
    d.y = __builtin_amdgcn_perm(oG[2], oB[2], pair_hi) | __builtin_amdgcn_perm(oR[1], oG[1], pair);
    d.z = __builtin_amdgcn_perm(oR[3], oG[3], pair_hi) | __builtin_amdgcn_perm(oB[3], oR[2], pair);
    return d;
}

// ... whichever layout the footprint's window has (wave-uniform)
__device__ __forceinline__ uint3 gather_blend_window(bool compact, const uint32_t (&bx)[4], const uint32_t (&by)[4], uint32_t lds_origin)
{
#ifndef MF_NO_COMPACT
    if (compact) return gather_blend_staged<MF_COMPACT_PITCH>(bx, by, lds_origin);
#endif
    (void)compact;
    return gather_blend_staged<LDS_PITCH>(bx, by, lds_origin);
}

// The 2 x 2 taps of the lane's four pixels straight from the frame (two unaligned 8-byte loads per pixel), for footprints without a
// staged window: a[j] = B0 G0 R0 B1 | G1 R1 . . of row iy (pixel ix, pixel ix+1), b[j] the same of row iy + 1.
__device__ __forceinline__ void gather_global(const uint32_t (&bx)[4], const uint32_t (&by)[4], const uint8_t* __restrict__ src, int W, uint2 (&a)[4], uint2 (&b)[4])
{
    const uint8_t* __restrict__ src1 = src + 3u * (uint32_t)W;   // row iy + 1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // ix = sx >> 5 = bits[5..21] (0x4B400000 >> 5 has no low 17 bits), same for iy
        const uint32_t t = umad24(__builtin_amdgcn_ubfe(by[j], 5, 17), (uint32_t)W, __builtin_amdgcn_ubfe(bx[j], 5, 17));
        const uint32_t o = t + (t << 1);
        __builtin_memcpy(&a[j], src + o, 8);
        __builtin_memcpy(&b[j], src1 + o, 8);
    }
}

// cv2.remap's bilinear blend (integer, 1/32-pixel weights) of the lane's four pixels from gather_global's layout: the 12 output bytes
// B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3.
__device__ __forceinline__ uint3 blend(const uint32_t (&bx)[4], const uint32_t (&by)[4], const uint2 (&a)[4], const uint2 (&b)[4])
{
    uint3 d;
    uint32_t oB[4], oG[4], oR[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // a[j].x = B0 G0 R0 B1, a[j].y = G1 R1 . .   (pixel ix, pixel ix+1 of row iy; b: row iy+1)
        // per channel the two horizontal neighbours side by side in 16-bit fields: X0 | X1 << 16
        const uint32_t Ba = __builtin_amdgcn_perm(a[j].y, a[j].x, 0x0C030C00u), Bb = __builtin_amdgcn_perm(b[j].y, b[j].x, 0x0C030C00u);
        const uint32_t Ga = __builtin_amdgcn_perm(a[j].y, a[j].x, 0x0C040C01u), Gb = __builtin_amdgcn_perm(b[j].y, b[j].x, 0x0C040C01u);
        const uint32_t Ra = __builtin_amdgcn_perm(a[j].y, a[j].x, 0x0C050C02u), Rb = __builtin_amdgcn_perm(b[j].y, b[j].x, 0x0C050C02u);
        // vertical lerp of both fields at once (each <= 255 * 32: no carry between them)
        const uint32_t fy = by[j] & 31u, wy = 32u - fy;
        const uint32_t vB = umad24(Bb, fy, __umul24(Ba, wy));
        const uint32_t vG = umad24(Gb, fy, __umul24(Ga, wy));
        const uint32_t vR = umad24(Rb, fy, __umul24(Ra, wy));
        // horizontal lerp: v_dot2_u32_u16 with the weight pair (32 - fx, fx) scaled by 64, so that ((sum + 512) >> 10)
        // lands in byte 2:  (sum + 512) * 64 < 2^24
        const uint32_t fx = bx[j] & 31u;
        const uint32_t wq = umad24(fx, 0x3FFFC0u, 2048u);       // 64 (32 - fx) | 64 fx << 16
        oB[j] = udot2(vB, wq, 32768u);
        oG[j] = udot2(vG, wq, 32768u);
        oR[j] = udot2(vR, wq, 32768u);
    }
    // the 12 result bytes sit in byte 2 of the 12 sums: 6 v_perm_b32 + 3 v_or_b32 gather them into B0 G0 R0 B1 | G1 R1 B2 G2 |
    // R2 B3 G3 R3  (pair = byte 2 of `lo` then byte 2 of `hi` in the two low bytes, zeros above)
    const uint32_t pair = 0x0C0C0602u;
    const uint32_t pair_hi = 0x06020C0Cu;                         // the same pair in the two high bytes: v_or joins them
    d.x = __builtin_amdgcn_perm(oB[1], oR[0], pair_hi) | __builtin_amdgcn_perm(oG[0], oB[0], pair);
    d.y = __builtin_amdgcn_perm(oG[2], oB[2], pair_hi) | __builtin_amdgcn_perm(oR[1], oG[1], pair);
    d.z = __builtin_amdgcn_perm(oR[3], oG[3], pair_hi) | __builtin_amdgcn_perm(oB[3], oR[2], pair);
    return d;
}


// ---- EXPERIMENT builds only (tools/phase_profile.sh; nothing of this is in the product library) ------------------------------------
// -DMF_EXP_SKIP=mask: TIMING-ONLY kernels in which the wavefronts of a path class return right after the plan test that selects
// them (their output is garbage): 1 hot, 2 border, 4 pair, 8 multi, 16 everything else, 32 every wavefront right after the plan and
// region words have arrived (no window copy), 64 every wavefront right after the window copy is issued.  What a class costs = the
// product kernel's time minus the time with that class skipped.
// -DMF_EXP_PHASES: every 64th HOT wavefront adds the s_memtime ticks between five points of its life to mf_exp_phase[] (read back by
// mf_exp_phase_read): entry -> plan + region arrived -> coordinates done (window copy in flight) -> window landed -> blend done.
// -DMF_EXP_TWICE: TIMING-ONLY: a hot wavefront does its per-pixel work twice (how much of a wavefront's cost is per wavefront?).
#ifndef MF_EXP_SKIP
#define MF_EXP_SKIP 0
#endif
#ifdef MF_EXP_PHASES
__device__ unsigned long long mf_exp_phase[8];
#define MF_EXP_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define MF_EXP_STAMP(var)
#endif

// STAGE_OK: the clip is 4-byte aligned, so the plan's STAGED windows can be copied by 16-byte global->LDS loads (always the case
// for buffers from hipMalloc / torch; the other instantiation ignores the windows).
// SCAN: the crop-boundary scan ALONE (crop_scan_kernel below): the same ownership and coordinate code for footprint t of frame f,
// then only the four edge tests of mfs.py:1075-1098 -- no window, no taps, no blend, no store.  The certified paths (hot, pair,
// multi) are compiled out: their footprints cannot set a crop flag (MF_REGION_DEEP / MF_REGION_NOFLAG) and are never handed in.
template <bool STAGE_OK, bool SCAN>
__device__ __forceinline__ void footprint_body(const uint32_t f, const uint32_t t, const FootPlan* __restrict__ plan, const FootRegion* __restrict__ regions,
                                               const WarpGeom& g, const uint8_t* __restrict__ frames,
                                               const double* __restrict__ records, uint8_t* __restrict__ out,
                                               const float* __restrict__ edges, int n, int W,
                                               int H, int C, uint32_t border, int32_t* __restrict__ crop, int32_t* __restrict__ clip)
{
    // inverse homographies of the footprint's candidate cells: [entry][Hi0..Hi8, pad] (80-byte rows)
    __shared__ __attribute__((aligned(16))) double s_hi[1][9][10];                // row 8: the "no cell" matrix, see OWN_NONE
    // source region of the footprint: MF_STAGE_ROWS rows of MF_STAGE_PITCH bytes (+ slack for the third dword of the last tap)
    __shared__ __attribute__((aligned(16))) uint8_t s_src_all[SCAN ? 16 : LDS_WINDOW_PAD + LDS_WINDOW_BYTES + 64];
    uint8_t* const s_src = &s_src_all[SCAN ? 0 : LDS_WINDOW_PAD];
    constexpr int wave = 0;
    MF_EXP_STAMP(exp_t0);
    const uint32_t ty = (__umulhi(t, g.div_m) + (t & g.div_pass)) >> g.div_s, tx = t - ty * g.nfx;
    const int xa = (int)(tx * (uint32_t)FOOT_W), ya = (int)(ty * (uint32_t)FOOT_H);
    const int lane = threadIdx.x;
#ifndef MF_NO_SPECMAT
    // SPECULATIVE matrix load.  A hot wavefront's life starts with three DEPENDENT scalar round trips -- kernel arguments, plan + region
    // words, the owner's inverse homography -- a quarter of its life (profiles/r05_phase_profile_cfg2.txt).  The owner of a hot footprint
    // is almost always the cell under the footprint's centre in the unwarped grid, which needs no plan: its matrix is requested HERE,
    // together with the plan words, and is there when they are.  The plan decides; a wrong guess (the neighbour cell owns the footprint,
    // or it is not hot at all) costs one unused 72-byte scalar load.  Inline asm: the compiler would sink the loads to their only use,
    // behind the plan's round trip; it does not know about them, so the hot path waits for them itself (spec_wait) before the first use.
    typedef uint32_t spec16_t __attribute__((ext_vector_type(16)));
    typedef uint32_t spec2_t __attribute__((ext_vector_type(2)));
    spec16_t hg_lo;
    spec2_t hg_hi;
    // ONLY in the instantiation that has a hot path (STAGE_OK && !SCAN).  Anywhere else the registers would be dead right behind the asm
    // statement, the compiler would hand them to the plan words' loads two lines further down, and -- scalar loads return out of order
    // -- whichever load lands last would win: a footprint of a frame stack that is not 4-byte aligned (odd frame sizes cut into frame
    // ranges: warp_kernel<false>) then ran on a few bytes of some cell's matrix instead of its plan about once in 200 launches and left
    // rows unwritten (found by a sweep over mf_warp_clip_u8c3's chunkings at the end of round 5).
#ifndef MF_GUARD_SELFTEST
    // (... and not in the timing-only builds that compile the hot path out -- MF_EXP_SKIP & (1 | 32 | 64) --: the same hazard, found there as a GPU
    // memory fault and, from the disassembly alone, by tools/isa_guard.py)
    constexpr bool SPECULATE = STAGE_OK && !SCAN && !(MF_EXP_SKIP & (1 | 32 | 64));
#else       // (tests/test_isa_guard.py builds THIS on purpose -- round 5's bug, the load in the instantiation without a hot path -- to see the guard fail)
    constexpr bool SPECULATE = !SCAN;
#endif
    const uint32_t k_guess = !SPECULATE ? 0u : min(__umulhi((uint32_t)ya + FOOT_H / 2, g.cell_mul_y) * g.mesh_cols + __umulhi((uint32_t)xa + FOOT_W / 2, g.cell_mul_x), g.cell_last);
    if (SPECULATE) {
        const uint64_t gaddr = (uint64_t)(uintptr_t)records + ((uint64_t)f * g.rec_frame_bytes + (uint64_t)k_guess * (uint32_t)(MF_CELL_DOUBLES * sizeof(double)));
        static_assert(MF_CELL_OFF_HI * sizeof(double) == 0x48 && MF_CELL_DOUBLES * sizeof(double) == 256, "offsets in the asm below");
        asm volatile("s_load_dwordx16 %0, %2, 0x48\n\ts_load_dwordx2 %1, %2, 0x88" : "=&s"(hg_lo), "=&s"(hg_hi) : "s"(gaddr));     // (early clobber: the address pair is read by both loads)
    }
#endif
    // (Round 6, measured and dropped: RE-ENTRY -- the wavefront jumps back to the kernel's first instruction as the next virtual workgroup, 2 or 4
    // footprints per wavefront with the product's code per trip: half / three quarters of the dispatches and of the end-of-life store waits gone,
    // byte-identical, +-0 -- so neither the launch rate nor a wavefront's latency limits the kernel, profiles/r06_ab_reentry.txt;
    // and -- profiles/r06_ab_prefetch.txt, profiles/README.md: touching the window lines of the footprint this
    // block index takes one or two frames on, to have them in the XCD's L2: +12...24 %; testing t >= per_frame BEHIND the plan's loads so
    // that all kernel arguments arrive in one scalar round trip instead of two: +-0.)
    const uint32_t fp = f * g.per_frame + t;                              // the footprint's slot in plan / regions
    typedef const __attribute__((address_space(4))) uint32_t* cword_t;
    const cword_t pw = (cword_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(plan) + 16u * fp);
    const cword_t rw = (cword_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(regions) + 8u * fp);
    const uint4 pv = make_uint4(pw[0], pw[1], pw[2], pw[3]);             // wave-uniform: scalar loads
    typedef const __attribute__((address_space(4))) uint64_t* cword2_t;
    const uint64_t region = *(cword2_t)rw;                               // both words in one load (the second is needed right after the first)
    const uint32_t rg = (uint32_t)region, src_dwords = (uint32_t)(region >> 32);
    const uint8_t* __restrict__ src = frames + (uint64_t)f * g.frame_bytes;
    const bool staged = STAGE_OK && !SCAN && (rg & MF_REGION_STAGED) != 0;
    if (!SCAN && (MF_EXP_SKIP & 32)) { asm volatile("" :: "s"(pv.x), "s"(pv.y), "s"(pv.z), "s"(pv.w), "s"(rg), "s"(src_dwords)); return; }
#ifdef MF_EXP_PHASES
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(pv.x), "s"(pv.w), "s"(rg) : "memory");
    MF_EXP_STAMP(exp_t1);
#endif
    // Lane -> footprint row.  The byte taps are served per group of 32 lanes, bank = dword address mod 32, and the eight lanes of a
    // footprint row take every third bank.  With the wide window (pitch 160 bytes = 40 banks) the rows 0..3 of lanes 0-31 start 0, 8, 16,
    // 24 banks apart: no two lanes on one bank.  With the COMPACT window (pitch 112 bytes = 28 banks) rows 0 and 3 would collide on four
    // banks -- every tap instruction 3.5 instead of 1.8 LDS cycles (tools/ubench_lds_rowmap.hip) -- so there lanes 0-31 take rows 0, 2, 4, 6
    // (0, 24, 16, 8 banks apart) and lanes 32-63 rows 1, 3, 5, 7 (set where the COMPACT copy is issued: wave-uniform).  Every lane still
    // owns four pixels of ONE row.
    uint32_t row = (uint32_t)lane >> 3;
    if (staged) {
        // Source region -> LDS, asynchronously (global_load_lds: no VGPRs, no ds_write).  Two layouts, chosen by the plan:
        //   COMPACT (hot footprints whose taps fit 9 rows x 112 bytes: ~3/4 of them): ONE load, lane i fetches the i-th 16-byte
        //           chunk (7 chunks per row), which lands at LDS offset 16 i.  (Lane 63 fetches the first chunk of a tenth row: unused,
        //           inside the frame because the region is DEEP.)
        //   wide    (12 rows x 160 bytes): lane i fetches the i-th and (64+i)-th chunk (10 chunks per row) -> LDS 16 i, 1024 + 16 i.
        // chunk i sits at row i / P, byte 16 (i % P) of the window = byte (i / P) (row_bytes - 16 P) + 16 i from gbase;
        // uniform base + opaque 32-bit lane offset keeps the address arithmetic 32-bit (saddr + voffset form)
        const uint8_t* __restrict__ gbase = src + ((uint64_t)src_dwords << 2);
        const lds_bytes_t window = lds_ptr(&s_src[0]);
#ifndef MF_NO_BORDER
        if (rg & MF_REGION_BORDER) {
            // BORDER window: the 12 rows only (its last row may be the frame's last: there is no 13th to fetch) -- chunks 0..63, then 64..119
            uint32_t o0 = __umul24(((uint32_t)lane * 205u) >> 11, g.row_bytes - (uint32_t)MF_STAGE_PITCH) + ((uint32_t)lane << 4);
            uint32_t o1 = __umul24((((uint32_t)lane + 64u) * 205u) >> 11, g.row_bytes - (uint32_t)MF_STAGE_PITCH) + (((uint32_t)lane << 4) + 1024u);
            asm("" : "+v"(o0));
            asm("" : "+v"(o1));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o0), (__attribute__((address_space(3))) void*)window, 16, 0, 0);
            if (lane < MF_STAGE_ROWS * 10 - 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o1), (__attribute__((address_space(3))) void*)(window + 1024), 16, 0, 0);
        } else
#endif
#ifndef MF_NO_COMPACT
        if (rg & MF_REGION_COMPACT) {
            uint32_t o0 = __umul24(((uint32_t)lane * 37u) >> 8, g.row_bytes - (uint32_t)MF_COMPACT_PITCH) + ((uint32_t)lane << 4);
            asm("" : "+v"(o0));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o0), (__attribute__((address_space(3))) void*)window, 16, 0, 0);
#ifndef MF_NO_ROWMAP
            row = (((uint32_t)lane >> 2) & 6u) | ((uint32_t)lane >> 5);
#endif
        } else
#endif
        {
        uint32_t o0 = __umul24(((uint32_t)lane * 205u) >> 11, g.row_bytes - (uint32_t)MF_STAGE_PITCH) + ((uint32_t)lane << 4);
        uint32_t o1 = __umul24((((uint32_t)lane + 64u) * 205u) >> 11, g.row_bytes - (uint32_t)MF_STAGE_PITCH) +
                      (((uint32_t)lane << 4) + 1024u);
        asm("" : "+v"(o0));
        asm("" : "+v"(o1));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o0), (__attribute__((address_space(3))) void*)window, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + o1), (__attribute__((address_space(3))) void*)(window + 1024), 16, 0, 0);
        }
    }
    // (a wavefront must not END with its global->LDS copy in flight: on this stack that is a GPU memory access fault -- tools/phase_variant_check.py --
    // so the timing-only returns below wait for it; `MF_EXP_RETURN` = that wait + return)
#define MF_EXP_RETURN do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; } while (0)
    if (!SCAN && (MF_EXP_SKIP & 64)) { asm volatile("" :: "s"(pv.x), "s"(pv.y), "s"(pv.z), "s"(pv.w)); MF_EXP_RETURN; }
    // taps are addressed by absolute LDS byte address (= LDS_PITCH iy + 3 ix - lds_origin): the window base is folded in
    const uint32_t lds_origin = (rg & MF_REGION_ORIGIN_MASK) - (uint32_t)(uintptr_t)&s_src[0];
    const crec_t frec = (crec_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(records) + f * g.rec_frame_bytes);
    const bool compact = STAGE_OK && !SCAN && (rg & MF_REGION_COMPACT) != 0;
    const int y = ya + (int)row;
    const int x0 = xa + (lane & 7) * 4;                                  // first of this lane's 4 pixels
    const double xs0 = (double)x0, yy = (double)y;

    if (STAGE_OK && !SCAN && (pv.x & (MF_PLAN_HOT << 16)) != 0) {
        // The plan certifies everything (~2/3 of the footprints at config-2 geometry): ONE cell owns all 256 pixels, its
        // denominator allows the trimmed reciprocal (UNIT), the footprint lies inside the frame, its window is staged and every
        // tap is at least two pixels inside the frame (DEEP: no crop flag either).  Straight-line code, all lanes active.
        if (MF_EXP_SKIP & 1) MF_EXP_RETURN;
#ifdef MF_EXP_TWICE
        {   // the per-pixel work a second time (rows + 8), stored over the first result
            float u2[4], v2[4];
            if (!cell_coords_fast(frec + (pv.x & 0xFFFu) * MF_CELL_DOUBLES, xs0, yy + 8.0, u2, v2))
                cell_coords<false>(frec + (pv.x & 0xFFFu) * MF_CELL_DOUBLES, xs0, yy + 8.0, x0, 0xFu, u2, v2, true);
            uint32_t bx2[4], by2[4];
            fixed_point(u2, v2, bx2, by2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint3 d2 = (rg & MF_REGION_COMPACT) ? gather_blend_staged<MF_COMPACT_PITCH>(bx2, by2, lds_origin) : gather_blend_staged(bx2, by2, lds_origin);
            uint8_t* __restrict__ dst2 = out + (uint64_t)f * g.frame_bytes;
            *reinterpret_cast<uint3*>(dst2 + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d2;
        }
#endif
        float u[4], v[4];
#ifndef MF_NO_FAST64
#ifndef MF_NO_SPECMAT
        bool have_coords = false;
        if ((pv.x >> 16) & MF_PLAN_FAST64) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hg_lo), "+s"(hg_hi));      // (the speculative load: long there -- the plan words came behind it)
            double Hi[9];
#pragma unroll
            for (int i = 0; i < 8; ++i) Hi[i] = __hiloint2double((int)hg_lo[2 * i + 1], (int)hg_lo[2 * i]);
            Hi[8] = __hiloint2double((int)hg_hi[1], (int)hg_hi[0]);
            if ((pv.x & 0xFFFu) != k_guess) {                             // (wave-uniform: the guess was the wrong cell)
                const crec_t rec = frec + (pv.x & 0xFFFu) * MF_CELL_DOUBLES;
#pragma unroll
                for (int i = 0; i < 9; ++i) Hi[i] = rec[MF_CELL_OFF_HI + i];
            }
            have_coords = coords_fast(Hi, xs0, yy, u, v);
        }
        if (!have_coords)
#else
        if (!((pv.x >> 16) & MF_PLAN_FAST64) || !cell_coords_fast(frec + (pv.x & 0xFFFu) * MF_CELL_DOUBLES, xs0, yy, u, v))
#endif
#endif
        cell_coords<false>(frec + (pv.x & 0xFFFu) * MF_CELL_DOUBLES, xs0, yy, x0, 0xFu, u, v, true);
        uint32_t bx[4], by[4];
        fixed_point(u, v, bx, by);
#ifdef MF_EXP_PHASES
        asm volatile("" :: "v"(bx[0]), "v"(bx[1]), "v"(bx[2]), "v"(bx[3]), "v"(by[0]), "v"(by[1]), "v"(by[2]), "v"(by[3]));
        MF_EXP_STAMP(exp_t2);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the window has landed in LDS
        MF_EXP_STAMP(exp_t3);
        uint8_t* __restrict__ dst = out + (uint64_t)f * g.frame_bytes;
        const uint3 d = gather_blend_window(compact, bx, by, lds_origin);
#ifdef MF_EXP_PHASES
        asm volatile("" :: "v"(d.x), "v"(d.y), "v"(d.z));
        MF_EXP_STAMP(exp_t4);
        // (the store, then the wait s_endpgm performs anyway -- a wavefront ends only when its store is acknowledged --, stamped: the TAIL of its life)
        *reinterpret_cast<uint3*>(dst + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MF_EXP_STAMP(exp_t5);
        if ((fp & 63u) == 0u && lane == 0) {
            atomicAdd(&mf_exp_phase[0], 1ull);
            atomicAdd(&mf_exp_phase[1], exp_t1 - exp_t0);
            atomicAdd(&mf_exp_phase[2], exp_t2 - exp_t1);
            atomicAdd(&mf_exp_phase[3], exp_t3 - exp_t2);
            atomicAdd(&mf_exp_phase[4], exp_t4 - exp_t3);
            atomicAdd(&mf_exp_phase[5], exp_t5 - exp_t4);
        }
        return;
#endif
        *reinterpret_cast<uint3*>(dst + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d;     // (STAGED implies W % 4 == 0)
        return;
    }

    const cedge_t fedge = (cedge_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(edges) + f * g.edge_frame_bytes);
#ifndef MF_NO_BORDER
    if (STAGE_OK && !SCAN && ((pv.x >> 16) & (MF_PLAN_VALID | MF_PLAN_BORDER)) == MF_PLAN_BORDER) {
        // BORDER path (the ring of footprints along the frame border of a stabilised clip, and the odd footprint a single cell only partly
        // covers: ~3 %): ONE candidate cell -- IN, or MIXED with one or two coded mask edges -- with a certified denominator; whole
        // footprint; every tap of a covered pixel lies in the staged window or on the ring of pixels just outside the frame, which is
        // painted into the window in the border colour here.  So the taps come from the staged gather like everywhere else: no
        // clamping, no per-tap selects (cv2.remap BORDER_CONSTANT, mfs.py:1063-1069).  Pixels the cell does not cover get the border
        // colour (the map template's (W+1, H+1), mfs.py:983-984) and take no part in the crop scan; a pixel inside the float32 error
        // band of an edge sends the wavefront to the general code.
        if (MF_EXP_SKIP & 2) MF_EXP_RETURN;
        const uint32_t k0 = pv.x & 0xFFFu;
        uint32_t cov = 0xFu;
        bool decided = true;
        if (!(pv.x & MF_PLAN_IN)) {
            const uint32_t cd = pv.z & 0x3Fu;
            const cedge_t ed = fedge + k0 * MF_EDGE_FLOATS;
            const cedge_t e1 = ed + 3u * (cd & 3u);
            const cedge_t e2 = ed + 3u * ((cd & 8u) ? ((cd >> 4) & 3u) : (cd & 3u));     // one-edge code: the same edge twice
            const float yf = (float)y, xf0 = (float)x0;
            const float r1 = __builtin_fmaf(e1[1], yf, e1[2]), r2 = __builtin_fmaf(e2[1], yf, e2[2]);
            float near = 1e30f;
            cov = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xf = xf0 + (float)j;
                const float gq = fminf(__builtin_fmaf(e1[0], xf, r1), __builtin_fmaf(e2[0], xf, r2));
                cov |= gq > EDGE_BAND ? (1u << j) : 0u;
                near = fminf(near, fabsf(gq));
            }
            decided = __ballot(!(near > EDGE_BAND)) == 0;           // (NaN coefficients: undecided)
        }
        if (decided) {
            float u[4], v[4];
            cell_coords<false>(frec + k0 * MF_CELL_DOUBLES, xs0, yy, x0, 0xFu, u, v, true);
            // crop-boundary scan of the covered pixels, mfs.py:1075-1098 (exact: Sterbenz, as on the generic path)
            {
                const float fWm1 = (float)(W - 1), fHm1 = (float)(H - 1);
                int c_left = 0, c_top = 0, c_right = W - 1, c_bottom = H - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((cov >> j) & 1u) {
                        const int x = x0 + j;
                        if (fabsf(u[j]) < 1.0f) c_left = max(c_left, x);
                        if (fabsf(u[j] - fWm1) < 1.0f) c_right = min(c_right, x);
                        if (fabsf(v[j]) < 1.0f) c_top = max(c_top, y);
                        if (fabsf(v[j] - fHm1) < 1.0f) c_bottom = min(c_bottom, y);
                    }
                }
                const bool any = c_left != 0 || c_top != 0 || c_right != W - 1 || c_bottom != H - 1;
                if (__ballot(any) != 0) {
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        c_left = max(c_left, __shfl_xor(c_left, off));
                        c_top = max(c_top, __shfl_xor(c_top, off));
                        c_right = min(c_right, __shfl_xor(c_right, off));
                        c_bottom = min(c_bottom, __shfl_xor(c_bottom, off));
                    }
                    if (lane == 0) {
                        // (per frame, mfs.py:1075-1098, and straight into the clip-level rectangle, mfs.py:1103-1106)
                        if (c_left != 0) { atomicMax(&crop[4 * f + 0], c_left); atomicMax(&clip[0], c_left); }
                        if (c_top != 0) { atomicMax(&crop[4 * f + 1], c_top); atomicMax(&clip[1], c_top); }
                        if (c_right != W - 1) { atomicMin(&crop[4 * f + 2], c_right); atomicMin(&clip[2], c_right); }
                        if (c_bottom != H - 1) { atomicMin(&crop[4 * f + 3], c_bottom); atomicMin(&clip[3], c_bottom); }
                    }
                }
            }
            uint32_t bx[4], by[4];
            fixed_point(u, v, bx, by);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the window has landed in LDS
            // Paint what lies just outside the frame: whole pixels (B, G, R) at the LDS address the gather will form for them -- tap
            // (ix, iy) sits at LDS_PITCH iy + 3 ix - lds_origin.  Column -1 / W for the rows -1 .. 12 of the window (14 lanes each), row
            // -1 / H for the columns that lie completely inside a window row (at most 53 lanes; no tap needs any other).  LDS operations of a wavefront execute
            // in order: the gather below sees these bytes.
            if (rg & (MF_REGION_PAINT_LEFT | MF_REGION_PAINT_RIGHT | MF_REGION_PAINT_TOP | MF_REGION_PAINT_BOTTOM)) {
                // (window origin in the frame from its first dword: row sy0, byte bs of the row -- bs can exceed the LDS pitch, so the
                // LDS origin does not split uniquely)
                const uint32_t first = src_dwords << 2;
                const int sy0 = (int)(first / g.row_bytes), bs = (int)(first - (uint32_t)sy0 * g.row_bytes);
                const int col0 = (bs + 2) / 3;                           // first column that starts inside the window's rows
                const auto paint = [&](int ix, int iy) {
                    const uint32_t at = (uint32_t)LDS_PITCH * (uint32_t)iy + 3u * (uint32_t)ix - lds_origin;      // (mod 2^32, like tap_address)
                    volatile __attribute__((address_space(3))) uint8_t* t = (volatile __attribute__((address_space(3))) uint8_t*)(uintptr_t)at;
                    t[0] = (uint8_t)border; t[1] = (uint8_t)(border >> 8); t[2] = (uint8_t)(border >> 16);
                };
                // (only the columns whose three bytes lie inside the LDS row: a neighbour's would land on the last bytes of the row in front or
                // the first of the row behind, which may be needed)
                const int ncols = (bs + LDS_PITCH - 3) / 3 - col0 + 1;
                if ((rg & MF_REGION_PAINT_TOP) && lane < ncols) paint(col0 + lane, -1);
                if ((rg & MF_REGION_PAINT_BOTTOM) && lane < ncols) paint(col0 + lane, H);
                // (the columns LAST: column -1 of a row shares its bytes with the end of the LDS row in front -- column 52, which no tap
                // needs -- and column W with the start of the row behind; the row paints above reach into both)
                if ((rg & MF_REGION_PAINT_LEFT) && lane < 14) paint(-1, sy0 - 1 + lane);
                if ((rg & MF_REGION_PAINT_RIGHT) && lane < 14) paint(W, sy0 - 1 + lane);
                __builtin_amdgcn_wave_barrier();
            }
            uint3 d = gather_blend_staged(bx, by, lds_origin);
            if (cov != 0xFu) {
                // pixels the cell does not cover: the border colour.  The lane's 12 bytes are B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3.
                const uint32_t b0 = border & 0xFFu, b1 = (border >> 8) & 0xFFu, b2 = (border >> 16) & 0xFFu;
                const uint32_t w0 = b0 | b1 << 8 | b2 << 16 | b0 << 24, w1 = b1 | b2 << 8 | b0 << 16 | b1 << 24, w2 = b2 | b0 << 8 | b1 << 16 | b2 << 24;
                const uint32_t m0 = ((cov & 1u) ? 0x00FFFFFFu : 0u) | ((cov & 2u) ? 0xFF000000u : 0u);
                const uint32_t m1 = ((cov & 2u) ? 0x0000FFFFu : 0u) | ((cov & 4u) ? 0xFFFF0000u : 0u);
                const uint32_t m2 = ((cov & 4u) ? 0x000000FFu : 0u) | ((cov & 8u) ? 0xFFFFFF00u : 0u);
                d.x = (d.x & m0) | (w0 & ~m0);
                d.y = (d.y & m1) | (w1 & ~m1);
                d.z = (d.z & m2) | (w2 & ~m2);
            }
            uint8_t* __restrict__ dstb = out + (uint64_t)f * g.frame_bytes;
            *reinterpret_cast<uint3*>(dstb + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d;     // (STAGED implies W % 4 == 0; the footprint is whole)
            return;
        }
    }
#endif
    if (STAGE_OK && !SCAN && (pv.y & MF_PLAN_HOT) != 0) {
        // Two cells share the footprint and the plan certifies the rest (a quarter of the footprints at config-2 geometry, 45 % at
        // config 3): the later cell wins wherever ONE of its mask edges passes -- one float32 fma per pixel -- and the other cell
        // owns what is left; denominators, window and interior as on the hot path.  Both inverse homographies go to LDS by
        // global->LDS DMA (one 80-byte load per cell, scalar base address), and every pixel reads its owner's row.
        if (MF_EXP_SKIP & 4) MF_EXP_RETURN;
        const uint32_t k0 = pv.x & 0xFFFu, k1 = (pv.x >> 16) & 0xFFFu;
        if (lane < 20) {
            uint32_t lo4 = (uint32_t)lane << 2;
            asm("" : "+v"(lo4));                        // (opaque: keeps the scalar base + 32-bit lane offset addressing form)
            uint64_t b0 = (uint64_t)(uintptr_t)(frec + k0 * MF_CELL_DOUBLES + MF_CELL_OFF_HI), b1 = (uint64_t)(uintptr_t)(frec + k1 * MF_CELL_DOUBLES + MF_CELL_OFF_HI);
            asm("" : "+s"(b0), "+s"(b1));                   // (whole bases in scalar registers: scalar base + 32-bit lane offset addressing)
            const uint8_t* __restrict__ g0 = (const uint8_t*)(uintptr_t)b0;
            const uint8_t* __restrict__ g1 = (const uint8_t*)(uintptr_t)b1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g0 + lo4),
                                             (__attribute__((address_space(3))) void*)lds_ptr(&s_hi[0][0][0]), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g1 + lo4),
                                             (__attribute__((address_space(3))) void*)lds_ptr(&s_hi[0][1][0]), 4, 0, 0);
        }
        const cedge_t eb = fedge + k0 * MF_EDGE_FLOATS + 3u * (pv.z & 3u);
#ifndef MF_NO_FASTPAIR
        if (pv.y & MF_PLAN_PAIR_FAST) {
            // LANE-UNIFORM form.  The edge crosses the footprint, but hardly ever the four pixels of a LANE when the lane's pixels run
            // ALONG it: for a mostly vertical edge (MF_PLAN_PAIR_VERT) the lanes are transposed -- lane l = column l % 32, rows
            // 4 (l / 32) .. + 3 -- for a mostly horizontal one they stay as they are (4 pixels of a row).  When every lane's four
            // pixels have ONE owner (wave-uniform test) the lane reads that owner's matrix from LDS once and runs the hot path's
            // cheap coordinate chain on it (one reciprocal per lane, plan-certified premises for both cells, midpoint guard): 5 LDS
            // matrix reads instead of 20 and 60 float64 operations instead of 100 per lane.  A transposed lane's pixels go back
            // through LDS (the window is no longer needed) to the row-major lanes that store them, 12 bytes each.
            // (two instantiations, chosen by a scalar branch: no per-lane selects on the wave-uniform direction)
            const auto lane_uniform = [&](auto vert_c) -> bool {
                constexpr bool VERT = decltype(vert_c)::value;
                const int px = VERT ? xa + (lane & 31) : x0, py = VERT ? ya + 4 * (lane >> 5) : y;
                // The edge function is affine along the lane, so its values at the lane's first and last pixel decide for all four: both
                // beyond the error band on the same side = one owner (the same evaluation as below -- a x + (b y + c), two fma -- so the
                // scaled band keeps its meaning: beyond +-1 the sign is the exact function's)
                const float g0 = __builtin_fmaf(eb[0], (float)px, __builtin_fmaf(eb[1], (float)py, eb[2]));
                const float g3 = VERT ? __builtin_fmaf(eb[0], (float)px, __builtin_fmaf(eb[1], (float)(py + 3), eb[2]))
                                      : __builtin_fmaf(eb[0], (float)(px + 3), __builtin_fmaf(eb[1], (float)py, eb[2]));
                const float lo = fminf(g0, g3), hi = fmaxf(g0, g3);
                const bool first = lo > EDGE_BAND, second = hi < -EDGE_BAND;         // (NaN coefficients: neither)
                if (__ballot(!(first || second)) != 0) return false;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // matrices and window have landed in LDS
                typedef const __attribute__((address_space(3))) double* lds_d;
                uint32_t hrow = (uint32_t)(uintptr_t)&s_hi[0][0][0] + (first ? 0u : OWN_ROW);
                asm("" : "+v"(hrow));                                    // (one address register + immediate offsets, not a select per load)
                const lds_d hp = (lds_d)(uintptr_t)hrow;
                const double Hl[9] = { hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7], hp[8] };
                float u[4], v[4];
                if (__ballot(coords_fast_dir<VERT>(Hl, (double)px, (double)py, u, v) < FAST64_NEAR) != 0) return false;
                uint32_t bx[4], by[4];
                fixed_point(u, v, bx, by);
                uint8_t* __restrict__ dst = out + (uint64_t)f * g.frame_bytes;
                uint3 d;
                if (VERT) {
                    uint32_t oB[4], oG[4], oR[4];
                    if (compact) gather_blend_sums<MF_COMPACT_PITCH>(bx, by, lds_origin, oB, oG, oR);
                    else gather_blend_sums<LDS_PITCH>(bx, by, lds_origin, oB, oG, oR);
                    // pixel (column c, row r) as B | G << 8 | R << 16 at word r * 32 + c of the (spent) window buffer ...
                    volatile uint32_t* tw = reinterpret_cast<volatile uint32_t*>(&s_src[0]);
                    const uint32_t at = (uint32_t)(4 * (lane >> 5)) * 32u + (uint32_t)(lane & 31);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        tw[at + 32u * (uint32_t)j] = __builtin_amdgcn_perm(oR[j], __builtin_amdgcn_perm(oG[j], oB[j], 0x0C0C0602u), 0x0C060100u);
                    __builtin_amdgcn_wave_barrier();
                    // ... and every lane takes the four pixels it stores: words 32 row + 4 (l % 8) .. + 3
                    // (its row is y - ya: the lane -> row mapping of the window layout, whatever it is)
                    const uint32_t w0 = 32u * (uint32_t)(y - ya) + 4u * ((uint32_t)lane & 7u);
                    const uint32_t p0 = tw[w0], p1 = tw[w0 + 1], p2 = tw[w0 + 2], p3 = tw[w0 + 3];
                    d.x = p0 | (p1 << 24);
                    d.y = (p1 >> 8) | (p2 << 16);
                    d.z = (p2 >> 16) | (p3 << 8);
                } else {
                    d = gather_blend_window(compact, bx, by, lds_origin);
                }
                *reinterpret_cast<uint3*>(dst + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d;
                return true;
            };
            if ((pv.y & MF_PLAN_PAIR_VERT) ? lane_uniform(std::true_type{}) : lane_uniform(std::false_type{})) return;
        }
#endif
        const float rb = __builtin_fmaf(eb[1], (float)y, eb[2]), xf0 = (float)x0;
        uint32_t own[4];
        float near = 1e30f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gb = __builtin_fmaf(eb[0], xf0 + (float)j, rb);
            own[j] = gb > EDGE_BAND ? 0u : OWN_ROW;
            near = fminf(near, fabsf(gb));
        }
        // (a pixel inside the float32 error band of the edge, or NaN coefficients: the general code below decides exactly)
        if (__ballot(!(near > EDGE_BAND)) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // matrices and window have landed in LDS
            float u[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double* hp = reinterpret_cast<const double*>(reinterpret_cast<const uint8_t*>(&s_hi[0][0][0]) + own[j]);
                const double2 h01 = *reinterpret_cast<const double2*>(hp), h23 = *reinterpret_cast<const double2*>(hp + 2);
                const double2 h45 = *reinterpret_cast<const double2*>(hp + 4), h67 = *reinterpret_cast<const double2*>(hp + 6);
                const double h8 = hp[8];
                const double xs = xs0 + (double)j;
                const double iw = recip_unit_range((xs * h67.x + yy * h67.y) + h8);
                u[j] = (float)(((xs * h01.x + yy * h01.y) + h23.x) * iw);
                v[j] = (float)(((xs * h23.y + yy * h45.x) + h45.y) * iw);
            }
            uint32_t bx[4], by[4];
            fixed_point(u, v, bx, by);
            uint8_t* __restrict__ dst = out + (uint64_t)f * g.frame_bytes;
            const uint3 d = gather_blend_window(compact, bx, by, lds_origin);
            *reinterpret_cast<uint3*>(dst + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d;
            return;
        }
    }

    if (STAGE_OK && !SCAN && (pv.z & MF_PLAN_HOT) != 0) {
        // Two to four cells, each MIXED one with one or two coded mask edges (the four cells around a mesh vertex, three of them, or a
        // pair the pair path did not take); window, interior and denominators certified, coverage not: a pixel that no listed cell
        // takes -- or one inside the float32 error band of an edge -- sends the wavefront to the general code.
        if (MF_EXP_SKIP & 8) MF_EXP_RETURN;
        const int ne = (int)((pv.z >> MF_PLAN_COUNT_SHIFT) & 3u) + 1;
        if (lane < 20) {
            uint32_t lo4 = (uint32_t)lane << 2;
            asm("" : "+v"(lo4));                        // (opaque: keeps the scalar base + 32-bit lane offset addressing form)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < ne) {
                    const uint32_t k = ((i < 2 ? pv.x : pv.y) >> (16 * (i & 1))) & 0xFFFu;
                    uint64_t bi = (uint64_t)(uintptr_t)(frec + k * MF_CELL_DOUBLES + MF_CELL_OFF_HI);
                    asm("" : "+s"(bi));
                    const uint8_t* __restrict__ gi = (const uint8_t*)(uintptr_t)bi;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gi + lo4),
                                                     (__attribute__((address_space(3))) void*)lds_ptr(&s_hi[0][i][0]), 4, 0, 0);
                }
            }
        }
        const float yf = (float)y, xf0 = (float)x0;
        uint32_t own[4] = { OWN_NONE, OWN_NONE, OWN_NONE, OWN_NONE };
        float near = 1e30f;
#pragma unroll
        for (int i = 3; i >= 0; --i) {                  // first entry last: it wins
            if (i < ne) {
                const uint32_t ent = (i < 2 ? pv.x : pv.y) >> (16 * (i & 1));
                if (ent & MF_PLAN_IN) {                 // (only the last entry can be IN: it owns what the others leave)
#pragma unroll
                    for (int j = 0; j < 4; ++j) own[j] = OWN_ROW * (uint32_t)i;
                } else {
                    const uint32_t cd = ((i < 2 ? pv.z : pv.w) >> (16 * (i & 1))) & 0x3Fu;
                    const cedge_t ed = fedge + (ent & 0xFFFu) * MF_EDGE_FLOATS;
                    const cedge_t e1 = ed + 3u * (cd & 3u);
                    const cedge_t e2 = ed + 3u * ((cd & 8u) ? ((cd >> 4) & 3u) : (cd & 3u));     // one-edge code: the same edge twice
                    const float r1 = __builtin_fmaf(e1[1], yf, e1[2]), r2 = __builtin_fmaf(e2[1], yf, e2[2]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xf = xf0 + (float)j;
                        const float gq = fminf(__builtin_fmaf(e1[0], xf, r1), __builtin_fmaf(e2[0], xf, r2));
                        own[j] = gq > EDGE_BAND ? OWN_ROW * (uint32_t)i : own[j];
                        near = fminf(near, fabsf(gq));
                    }
                }
            }
        }
        const uint32_t worst = max(max(own[0], own[1]), max(own[2], own[3]));
        if (__ballot(!(near > EDGE_BAND) || worst == OWN_NONE) == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // matrices and window have landed in LDS
            float u[4], v[4];
            bool have = false;
#ifndef MF_NO_FASTMULTI
            if (pv.z & MF_PLAN_MULTI_FAST) {
                // every listed cell satisfies the premises of the cheap chain: fused affine forms per pixel from its owner's matrix, one
                // reciprocal for the lane's four denominators, midpoint guard (a flagged wavefront takes the exact chain below)
                double wq[4], nq[4], mq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double* hp = reinterpret_cast<const double*>(reinterpret_cast<const uint8_t*>(&s_hi[0][0][0]) + own[j]);
                    const double2 h01 = *reinterpret_cast<const double2*>(hp), h23 = *reinterpret_cast<const double2*>(hp + 2);
                    const double2 h45 = *reinterpret_cast<const double2*>(hp + 4), h67 = *reinterpret_cast<const double2*>(hp + 6);
                    const double xs = xs0 + (double)j;
                    nq[j] = __builtin_fma(xs, h01.x, __builtin_fma(yy, h01.y, h23.x));
                    mq[j] = __builtin_fma(xs, h23.y, __builtin_fma(yy, h45.x, h45.y));
                    wq[j] = __builtin_fma(xs, h67.x, __builtin_fma(yy, h67.y, hp[8]));
                }
                have = __ballot(cheap_quotients(wq, nq, mq, u, v) < FAST64_NEAR) == 0;
            }
#endif
            if (!have) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double* hp = reinterpret_cast<const double*>(reinterpret_cast<const uint8_t*>(&s_hi[0][0][0]) + own[j]);
                    const double2 h01 = *reinterpret_cast<const double2*>(hp), h23 = *reinterpret_cast<const double2*>(hp + 2);
                    const double2 h45 = *reinterpret_cast<const double2*>(hp + 4), h67 = *reinterpret_cast<const double2*>(hp + 6);
                    const double h8 = hp[8];
                    const double xs = xs0 + (double)j;
                    const double iw = recip_unit_range((xs * h67.x + yy * h67.y) + h8);
                    u[j] = (float)(((xs * h01.x + yy * h01.y) + h23.x) * iw);
                    v[j] = (float)(((xs * h23.y + yy * h45.x) + h45.y) * iw);
                }
            }
            uint32_t bx[4], by[4];
            fixed_point(u, v, bx, by);
            uint8_t* __restrict__ dst = out + (uint64_t)f * g.frame_bytes;
            const uint3 d = gather_blend_window(compact, bx, by, lds_origin);
            *reinterpret_cast<uint3*>(dst + ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u) = d;
            return;
        }
    }

    // Everything else: more candidate cells, uncertified denominators, frame borders, uncovered pixels.
    if (!SCAN && (MF_EXP_SKIP & 16)) MF_EXP_RETURN;
    uint8_t* __restrict__ dst = out + (uint64_t)f * g.frame_bytes;
    const uint32_t limit = (int)f == n - 1 ? g.frame_bytes : 0xFFFFFFFFu;   // only the last frame has nothing behind it
    const float fWm1 = (float)(W - 1), fHm1 = (float)(H - 1);
    const bool fast_store = (W & 3) == 0;
    // active = y < H && x0 < W, built on the scalar unit as a lane mask (rows_in rows of the footprint, and in each the first
    // cols_in groups of four pixels, start inside the frame) and turned into the branch condition without a v_cmp
    const int rows_in = min(FOOT_H, H - ya), cols_in = min(FOOT_W / 4, (W - xa + 3) >> 2);
    const uint32_t row_bits = ((1u << cols_in) - 1u) * 0x01010101u;
    const uint64_t lanes_in = (((uint64_t)row_bits << 32) | row_bits) & (~0ull >> (64 - 8 * rows_in));
    const bool active = __builtin_amdgcn_inverse_ballot_w64(lanes_in);
    {
        // Source coordinates of the lane's 4 pixels; (W+1, H+1) = "no cell covers it" (mfs.py:983-984).
        float u[4], v[4];
        if ((pv.x & (MF_PLAN_IN | MF_PLAN_VALID)) == (MF_PLAN_IN | MF_PLAN_VALID) && (pv.w >> 16) != MF_PLAN_OVERFLOW) {
            // one cell owns the whole footprint (the common case): no per-pixel test, no merging
            cell_coords<false>(frec + (pv.x & 0xFFFu) * MF_CELL_DOUBLES, xs0, yy, x0, 0xFu, u, v, ((pv.x >> 16) & MF_PLAN_UNIT) != 0);
        } else if ((pv.w >> 16) == MF_PLAN_OVERFLOW) {
            // more than 8 candidate cells: test every cell of the recorded range, last cell first
            uint32_t unowned = 0;
            int Wp = W + 1, Hp = H + 1;                      // (opaque: keeps the conversions inside this rare branch)
            asm volatile("" : "+s"(Wp), "+s"(Hp));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u[j] = (float)Wp;
                v[j] = (float)Hp;
                if (x0 + j < W && y < H) unowned |= 1u << j;
            }
            const int r_lo = pv.x & 0xFFFF, c_lo = pv.y & 0xFFFF, c_hi = pv.y >> 16;
            int cr = pv.x >> 16, cc = c_hi;
            bool done = __ballot(unowned != 0) == 0;
            while (!done && cr >= r_lo) {
                const int k = cr * C + cc;
                if (--cc < c_lo) { cc = c_hi; --cr; }
                const crec_t rec = frec + (uint32_t)k * MF_CELL_DOUBLES;
                if (rec[MF_CELL_OFF_STATUS] != 0.0) continue;
                const uint32_t pass = cell_mask_test(rec, xs0, yy, x0, y, unowned);
                if (__ballot(pass != 0) == 0) continue;
                unowned &= ~pass;
                cell_coords<true>(rec, xs0, yy, x0, pass, u, v);
                done = __ballot(unowned != 0) == 0;
            }
        } else {
            // Several cells share the footprint.  (a) their inverse homographies go to LDS; (b) ownership is
            // resolved per pixel, last cell first: a float32 evaluation of the cell's four edge functions
            // decides unless the pixel is within the float32 error band of a mask edge, then the float64 test; (c) every
            // pixel computes its coordinates ONCE with its owner's matrix read from LDS.
            int ne = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t d = i < 2 ? pv.x : i < 4 ? pv.y : i < 6 ? pv.z : pv.w;
                if (ne == i && ((d >> (16 * (i & 1))) & MF_PLAN_VALID)) ne = i + 1;
            }
            // (a) entry e's nine doubles are 18 consecutive dwords of its record: lanes 0..19 copy them (and two dwords of padding)
            // straight into row e of s_hi, one global->LDS load per entry with a scalar base address -- no per-lane cell lookup
            if (lane < 20) {
                uint32_t lo4 = (uint32_t)lane << 2;
            asm("" : "+v"(lo4));                        // (opaque: keeps the scalar base + 32-bit lane offset addressing form)
#pragma unroll 1
                for (int e = 0; e < ne; ++e) {
                    const uint32_t d = e < 2 ? pv.x : e < 4 ? pv.y : e < 6 ? pv.z : pv.w;
                    const uint32_t k = (d >> (16 * (e & 1))) & 0xFFFu;
                    const uint8_t* __restrict__ g = (const uint8_t*)(uintptr_t)(frec + k * MF_CELL_DOUBLES + MF_CELL_OFF_HI);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lo4),
                                                     (__attribute__((address_space(3))) void*)lds_ptr(&s_hi[wave][e][0]), 4, 0, 0);
                }
            }
            uint32_t own[4];                                            // byte offset of the owner's matrix row (OWN_ROW * entry)
            if (!(rg & MF_REGION_DEEP) && lane < 10) {                  // only uncertified footprints can have uncovered pixels
                int Wp = W + 1, Hp = H + 1;                              // (opaque: keeps the conversions inside this branch)
                asm volatile("" : "+s"(Wp), "+s"(Hp));
                s_hi[wave][8][lane] = lane == 2 ? (double)Wp : lane == 5 ? (double)Hp : lane == 8 ? 1.0 : 0.0;
            }
            const float yf = (float)y, xf0 = (float)x0;
            // The common shape -- exactly two cells, each with ONE mask edge crossing the footprint (a footprint on the
            // border between two cells): one fma per pixel and cell decides, straight-line.
            const uint32_t cd0 = pv.z & 0x3Fu, cd1 = (pv.z >> 16) & 0x3Fu, cd2 = pv.w & 0x3Fu, cd3 = (pv.w >> 16) & 0x3Fu;
            const bool pair = ne == 2 && !(pv.x & MF_PLAN_IN) && !((pv.x >> 16) & MF_PLAN_IN) && cd0 < 4u && cd1 < 4u;
            // four cells around a mesh vertex, each with the two edges that meet there uncertain
            const bool quad = ne == 4 && !((pv.x | (pv.x >> 16) | pv.y | (pv.y >> 16)) & MF_PLAN_IN) &&
                              (cd0 & cd1 & cd2 & cd3 & 8u) != 0;
            bool general = !(pair || quad);
            if (quad) {
                float near = 1e30f;
#pragma unroll
                for (int j = 0; j < 4; ++j) own[j] = OWN_NONE;
#pragma unroll
                for (int i = 3; i >= 0; --i) {                  // first entry last: it wins
                    const uint32_t ent = (i < 2 ? pv.x : pv.y) >> (16 * (i & 1));
                    const uint32_t cd = i == 0 ? cd0 : i == 1 ? cd1 : i == 2 ? cd2 : cd3;
                    const cedge_t ed = fedge + (ent & 0xFFFu) * MF_EDGE_FLOATS;
                    const cedge_t e1 = ed + 3u * (cd & 3u);
                    const cedge_t e2 = ed + 3u * ((cd >> 4) & 3u);
                    const float r1 = __builtin_fmaf(e1[1], yf, e1[2]), r2 = __builtin_fmaf(e2[1], yf, e2[2]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xf = xf0 + (float)j;
                        const float g = fminf(__builtin_fmaf(e1[0], xf, r1), __builtin_fmaf(e2[0], xf, r2));
                        own[j] = g > EDGE_BAND ? OWN_ROW * (uint32_t)i : own[j];
                        near = fminf(near, fabsf(g));
                    }
                }
                general = __ballot(!(near > EDGE_BAND) && active) != 0;
            }
            if (pair) {
                const cedge_t eb = fedge + (pv.x & 0xFFFu) * MF_EDGE_FLOATS + 3u * cd0;            // later cell: wins
                const cedge_t ea = fedge + ((pv.x >> 16) & 0xFFFu) * MF_EDGE_FLOATS + 3u * cd1;
                const float rb = __builtin_fmaf(eb[1], yf, eb[2]), ra = __builtin_fmaf(ea[1], yf, ea[2]);
                float near = 1e30f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xf = xf0 + (float)j;
                    const float gb = __builtin_fmaf(eb[0], xf, rb), ga = __builtin_fmaf(ea[0], xf, ra);
                    own[j] = gb > EDGE_BAND ? 0u : (ga > EDGE_BAND ? OWN_ROW : OWN_NONE);
                    near = fminf(near, fminf(fabsf(gb), fabsf(ga)));
                }
                // a pixel inside the float32 error band of a mask edge (or NaN coefficients): the general path decides exactly
                general = __ballot(!(near > EDGE_BAND) && active) != 0;
            }
            if (general) {
            uint32_t unowned = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                own[j] = OWN_NONE;
                if (x0 + j < W && y < H) unowned |= 1u << j;
            }
            bool done = __ballot(unowned != 0) == 0;
#pragma unroll 1
            for (int i = 0; i < ne && !done; ++i) {
                const uint32_t d = i < 2 ? pv.x : i < 4 ? pv.y : i < 6 ? pv.z : pv.w;
                const uint32_t e = (d >> (16 * (i & 1))) & 0xFFFFu;
                const uint32_t k = e & 0xFFFu;
                uint32_t pass = unowned;                                   // IN: every unowned pixel passes
                if (!(e & MF_PLAN_IN)) {
                    const cedge_t ed = fedge + k * MF_EDGE_FLOATS;
                    // short lists carry an edge code: only one of the four edge functions can fail in this footprint
                    const uint32_t code = ne <= 4 ? (((i < 2 ? pv.z : pv.w) >> (16 * (i & 1))) & 0x3Fu) : 4u;
                    uint32_t ok = 0, amb = 0;
                    if (code < 4u) {
                        const cedge_t e1 = ed + 3u * code;
                        const float rr = __builtin_fmaf(e1[1], yf, e1[2]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float g = __builtin_fmaf(e1[0], xf0 + (float)j, rr);
                            ok |= g > EDGE_BAND ? (1u << j) : 0u;
                            amb |= ((g > EDGE_BAND) | (g < -EDGE_BAND)) ? 0u : (1u << j);
                        }
                    } else {
                        const float r0 = __builtin_fmaf(ed[1], yf, ed[2]), r1 = __builtin_fmaf(ed[4], yf, ed[5]);
                        const float r2 = __builtin_fmaf(ed[7], yf, ed[8]), r3 = __builtin_fmaf(ed[10], yf, ed[11]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xf = xf0 + (float)j;
                            const float g = fminf(fminf(__builtin_fmaf(ed[0], xf, r0), __builtin_fmaf(ed[3], xf, r1)),
                                                  fminf(__builtin_fmaf(ed[6], xf, r2), __builtin_fmaf(ed[9], xf, r3)));
                            ok |= g > EDGE_BAND ? (1u << j) : 0u;
                            amb |= ((g > EDGE_BAND) | (g < -EDGE_BAND)) ? 0u : (1u << j);     // NaN (irregular cell) -> ambiguous
                        }
                    }
                    amb &= unowned;
                    if (__ballot(amb != 0) != 0)
                        ok = (ok & ~amb) | cell_mask_test(frec + k * MF_CELL_DOUBLES, xs0, yy, x0, y, amb);
                    pass = ok & unowned;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) own[j] = ((pass >> j) & 1u) ? OWN_ROW * (uint32_t)i : own[j];
                unowned &= ~pass;
                done = __ballot(unowned != 0) == 0;
            }
            }
            // (c) coordinates, once per pixel, owner's matrix from LDS.  Optimistic: the
            // trimmed reciprocal is applied straight away (keeps one pixel's intermediates live instead of four) and the
            // rare footprint with a denominator outside [0.5, 2) is redone with the generic division.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the matrices (and the window, issued before them) have landed
            uint32_t eor = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double* hp = reinterpret_cast<const double*>(reinterpret_cast<const uint8_t*>(&s_hi[wave][0][0]) + own[j]);
                const double2 h01 = *reinterpret_cast<const double2*>(hp), h23 = *reinterpret_cast<const double2*>(hp + 2);
                const double2 h45 = *reinterpret_cast<const double2*>(hp + 4), h67 = *reinterpret_cast<const double2*>(hp + 6);
                const double h8 = hp[8];
                const double xs = xs0 + (double)j;
                const double w = (xs * h67.x + yy * h67.y) + h8;
                const double nx = (xs * h01.x + yy * h01.y) + h23.x;
                const double ny = (xs * h23.y + yy * h45.x) + h45.y;
                eor |= (uint32_t)__builtin_amdgcn_frexp_exp(w);
                const double iw = recip_unit_range(w);
                const float un = (float)(nx * iw), vn = (float)(ny * iw);
                u[j] = un;
                v[j] = vn;
            }
            if (__ballot(eor > 1u) != 0) {                             // far-from-affine cell: generic division
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double* hp = reinterpret_cast<const double*>(reinterpret_cast<const uint8_t*>(&s_hi[wave][0][0]) + own[j]);
                    const double xs = xs0 + (double)j;
                    const double w = (xs * hp[6] + yy * hp[7]) + hp[8];
                    const bool ok = fabs(w) > 1.1920928955078125e-07;
                    const double iw = 1.0 / w;
                    u[j] = ok ? (float)(((xs * hp[0] + yy * hp[1]) + hp[2]) * iw) : 0.0f;
                    v[j] = ok ? (float)(((xs * hp[3] + yy * hp[4]) + hp[5]) * iw) : 0.0f;
                }
            }
        }

        if (SCAN) {
            // The crop-boundary scan alone, mfs.py:1075-1098 (the same tests as on the generic path below; there they run only when
            // some pixel of the footprint is not deep inside the frame -- a pixel that is cannot pass any of them).
            int c_left = 0, c_top = 0, c_right = W - 1, c_bottom = H - 1;
            if (active) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = x0 + j;
                    if (x < W) {
                        if (fabsf(u[j]) < 1.0f) c_left = max(c_left, x);
                        if (fabsf(u[j] - fWm1) < 1.0f) c_right = min(c_right, x);
                        if (fabsf(v[j]) < 1.0f) c_top = max(c_top, y);
                        if (fabsf(v[j] - fHm1) < 1.0f) c_bottom = min(c_bottom, y);
                    }
                }
            }
            const bool any = c_left != 0 || c_top != 0 || c_right != W - 1 || c_bottom != H - 1;
            if (__ballot(any) != 0) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    c_left = max(c_left, __shfl_xor(c_left, off));
                    c_top = max(c_top, __shfl_xor(c_top, off));
                    c_right = min(c_right, __shfl_xor(c_right, off));
                    c_bottom = min(c_bottom, __shfl_xor(c_bottom, off));
                }
                if (lane == 0) {
                    // (per frame, mfs.py:1075-1098, and straight into the clip-level rectangle, mfs.py:1103-1106)
                    if (c_left != 0) { atomicMax(&crop[4 * f + 0], c_left); atomicMax(&clip[0], c_left); }
                    if (c_top != 0) { atomicMax(&crop[4 * f + 1], c_top); atomicMax(&clip[1], c_top); }
                    if (c_right != W - 1) { atomicMin(&crop[4 * f + 2], c_right); atomicMin(&clip[2], c_right); }
                    if (c_bottom != H - 1) { atomicMin(&crop[4 * f + 3], c_bottom); atomicMin(&clip[3], c_bottom); }
                }
            }
            return;
        }
        // cv2.remap: 1/32-pixel fixed point (round half to even), bilinear gather, store.
        uint32_t bx[4], by[4];
        fixed_point(u, v, bx, by);
        bool fast = true;
        if (!(staged && (rg & MF_REGION_DEEP))) {        // (DEEP: every pixel has an owner and every tap is deep inside: nothing to check)
            uint32_t dxm = 0, dym = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dxm = max(dxm, bx[j] - (0x4B400000u + 64u));
                dym = max(dym, by[j] - (0x4B400000u + 64u));
            }
            // "deep interior": 2 <= ix <= W-3 and 2 <= iy <= H-3 for all four pixels.  Then both taps in x and
            // y are inside the frame, the 8-byte loads stay inside the row, and no crop flag can be set
            // (u >= 2 - 1/64 and u < W - 2, same for v).
            // (a frame of fewer than five columns or rows has no such pixel: the bounds would wrap around as unsigned numbers)
            const bool deep = W >= 5 && H >= 5 && dxm <= (uint32_t)(32 * (W - 3) + 31 - 64) && dym <= (uint32_t)(32 * (H - 3) + 31 - 64);
            fast = __ballot(active && !deep) == 0;
        }
        uint3 d;                                                        // the lane's 12 output bytes
        if (staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window has landed in LDS
        if (fast) {
            // fast path (wave-uniform): every pixel of the footprint samples the deep interior
            if (active) {
                if (staged) {
                    d = gather_blend_window(compact, bx, by, lds_origin);
                } else {
                    uint2 a[4], b[4];
                    gather_global(bx, by, src, W, a, b);
                    d = blend(bx, by, a, b);
                }
            }
        } else {
            // generic path: frame borders, uncovered pixels, crop flags, out-of-range coordinates (3 % of a stabilised clip's
            // footprints -- the ring along the frame border whose pixels sample outside the frame)
            int c_left = 0, c_top = 0, c_right = W - 1, c_bottom = H - 1;
            if (active) {
            // sx = rint(32 u) sits in the low bits of fixed_point's raw floats while |sx| < 2^22; coordinates beyond that (a cell far
            // from affine) take cv2's own rounding with its saturation
            uint32_t spread = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                spread = max(spread, max(bx[j] - (0x4B400000u - 0x200000u), by[j] - (0x4B400000u - 0x200000u)));
            const bool narrow = __ballot(spread >= 0x400000u) == 0;
            const bool has_tail = limit != 0xFFFFFFFFu;      // last frame of the stack: the 4-byte load of its last pixel is shifted back
            uint32_t oB[4], oG[4], oR[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float uu = u[j], vv = v[j];
                const int x = x0 + j;
                // crop-boundary scan, mfs.py:1075-1098: |u - e| < 1.  The float32 differences are exact
                // whenever they are smaller than 1 in magnitude (Sterbenz), so the tests are exact.
                if (x < W) {                                 // (a lane's last pixels may lie beyond the frame when W % 4 != 0)
                    if (fabsf(uu) < 1.0f) c_left = max(c_left, x);
                    if (fabsf(uu - fWm1) < 1.0f) c_right = min(c_right, x);
                    if (fabsf(vv) < 1.0f) c_top = max(c_top, y);
                    if (fabsf(vv - fHm1) < 1.0f) c_bottom = min(c_bottom, y);
                }
                const int sxx = narrow ? (int)(bx[j] - 0x4B400000u) : cv_round_f32(uu * 32.0f);
                const int syy = narrow ? (int)(by[j] - 0x4B400000u) : cv_round_f32(vv * 32.0f);
                const int ix = sxx >> 5, iy = syy >> 5;      // (saturation to int16 cannot change any decision below)
                // The four taps, branch-free: each load goes to the position clamped into the frame and the tap is replaced by the
                // border colour afterwards when it lies outside (a 2 x 2 footprint outside altogether needs no special case: four
                // border-colour taps with weights that sum to 1024 give the border colour exactly).  All sixteen loads of the lane
                // are in flight together.
                const bool in_x0 = (unsigned)ix < (unsigned)W, in_x1 = (unsigned)(ix + 1) < (unsigned)W;
                const bool in_y0 = (unsigned)iy < (unsigned)H, in_y1 = (unsigned)(iy + 1) < (unsigned)H;
                const uint32_t cx0 = (uint32_t)min(max(ix, 0), W - 1), cx1 = (uint32_t)min(max(ix + 1, 0), W - 1);
                if (staged) {
                    // STAGED: the window holds every tap position CLAMPED into the frame (cell_table.hip), so the four taps are LDS
                    // byte loads (a pixel without owner sits at (W+1, H+1): its clamped position may lie outside the window --
                    // whatever the load returns is replaced by the border colour below).  (Unaligned 4-byte LDS loads instead:
                    // +2 % kernel time; two pixels' loads in flight: register spills.)
                    const uint32_t pitch = compact ? (uint32_t)MF_COMPACT_PITCH : (uint32_t)LDS_PITCH;      // (wave-uniform)
                    const uint32_t ra = umad24((uint32_t)min(max(iy, 0), H - 1), pitch, 0u - lds_origin);
                    const uint32_t rb = umad24((uint32_t)min(max(iy + 1, 0), H - 1), pitch, 0u - lds_origin);
                    TapRegs t;
                    taps_clamped(umad24(cx0, 3u, ra), umad24(cx1, 3u, ra), umad24(cx0, 3u, rb), umad24(cx1, 3u, rb), t);
                    const bool i00 = in_x0 && in_y0, i01 = in_x1 && in_y0, i10 = in_x0 && in_y1, i11 = in_x1 && in_y1;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint32_t bc = (border >> (8 * c)) & 0xFFu;
                        t.lo[c] = i00 ? t.lo[c] : bc;
                        t.hi[c] = i01 ? t.hi[c] : bc << 16;
                        t.lo[3 + c] = i10 ? t.lo[3 + c] : bc;
                        t.hi[3 + c] = i11 ? t.hi[3 + c] : bc << 16;
                    }
                    blend_pixel((uint32_t)sxx, (uint32_t)syy, t, oB[j], oG[j], oR[j]);
                    continue;
                }
                const uint32_t r0 = (uint32_t)min(max(iy, 0), H - 1) * (uint32_t)W, r1 = (uint32_t)min(max(iy + 1, 0), H - 1) * (uint32_t)W;
                const uint32_t o00 = (r0 + cx0) * 3u, o01 = (r0 + cx1) * 3u, o10 = (r1 + cx0) * 3u, o11 = (r1 + cx1) * 3u;
                uint32_t p00, p01, p10, p11;                 // B | G << 8 | R << 16 | (next byte) << 24
                if (has_tail) {
                    const uint32_t k00 = o00 + 4u > limit, k01 = o01 + 4u > limit, k10 = o10 + 4u > limit, k11 = o11 + 4u > limit;
                    __builtin_memcpy(&p00, src + (o00 - k00), 4); p00 >>= 8u * k00;
                    __builtin_memcpy(&p01, src + (o01 - k01), 4); p01 >>= 8u * k01;
                    __builtin_memcpy(&p10, src + (o10 - k10), 4); p10 >>= 8u * k10;
                    __builtin_memcpy(&p11, src + (o11 - k11), 4); p11 >>= 8u * k11;
                } else {
                    __builtin_memcpy(&p00, src + o00, 4);
                    __builtin_memcpy(&p01, src + o01, 4);
                    __builtin_memcpy(&p10, src + o10, 4);
                    __builtin_memcpy(&p11, src + o11, 4);
                }
                p00 = in_x0 && in_y0 ? p00 : border;
                p01 = in_x1 && in_y0 ? p01 : border;
                p10 = in_x0 && in_y1 ? p10 : border;
                p11 = in_x1 && in_y1 ? p11 : border;
                // the blend of the fast path: per channel the two horizontal neighbours in 16-bit fields, both lerped vertically at
                // once, then v_dot2_u32_u16 horizontally (byte 3 of the taps is never selected)
                const uint32_t fy = (uint32_t)syy & 31u, wy = 32u - fy;
                const uint32_t vB = umad24(__builtin_amdgcn_perm(p11, p10, 0x0C040C00u), fy, __umul24(__builtin_amdgcn_perm(p01, p00, 0x0C040C00u), wy));
                const uint32_t vG = umad24(__builtin_amdgcn_perm(p11, p10, 0x0C050C01u), fy, __umul24(__builtin_amdgcn_perm(p01, p00, 0x0C050C01u), wy));
                const uint32_t vR = umad24(__builtin_amdgcn_perm(p11, p10, 0x0C060C02u), fy, __umul24(__builtin_amdgcn_perm(p01, p00, 0x0C060C02u), wy));
                const uint32_t wq = umad24((uint32_t)sxx & 31u, 0x3FFFC0u, 2048u);          // 64 (32 - fx) | 64 fx << 16
                oB[j] = udot2(vB, wq, 32768u);
                oG[j] = udot2(vG, wq, 32768u);
                oR[j] = udot2(vR, wq, 32768u);
            }
            const uint32_t pair = 0x0C0C0602u, pair_hi = 0x06020C0Cu;       // byte 2 of each sum, as in gather_blend_staged
            d.x = __builtin_amdgcn_perm(oB[1], oR[0], pair_hi) | __builtin_amdgcn_perm(oG[0], oB[0], pair);
            d.y = __builtin_amdgcn_perm(oG[2], oB[2], pair_hi) | __builtin_amdgcn_perm(oR[1], oG[1], pair);
            d.z = __builtin_amdgcn_perm(oR[3], oG[3], pair_hi) | __builtin_amdgcn_perm(oB[3], oR[2], pair);
            }
            // Crop bounds (only this path can set one): wave reduction, then at most one atomic per bound and wavefront.
            const bool any = c_left != 0 || c_top != 0 || c_right != W - 1 || c_bottom != H - 1;
            if (__ballot(any) != 0) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    c_left = max(c_left, __shfl_xor(c_left, off));
                    c_top = max(c_top, __shfl_xor(c_top, off));
                    c_right = min(c_right, __shfl_xor(c_right, off));
                    c_bottom = min(c_bottom, __shfl_xor(c_bottom, off));
                }
                if (lane == 0) {
                    // (per frame, mfs.py:1075-1098, and straight into the clip-level rectangle, mfs.py:1103-1106)
                    if (c_left != 0) { atomicMax(&crop[4 * f + 0], c_left); atomicMax(&clip[0], c_left); }
                    if (c_top != 0) { atomicMax(&crop[4 * f + 1], c_top); atomicMax(&clip[1], c_top); }
                    if (c_right != W - 1) { atomicMin(&crop[4 * f + 2], c_right); atomicMin(&clip[2], c_right); }
                    if (c_bottom != H - 1) { atomicMin(&crop[4 * f + 3], c_bottom); atomicMin(&clip[3], c_bottom); }
                }
            }
        }
        if (active) {                                                   // the lane's 12 output bytes
            const uint32_t o = ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 3u;
            if (fast_store) {                                           // W % 4 == 0: an active lane's four pixels are all inside
                *reinterpret_cast<uint3*>(dst + o) = d;
            } else if (x0 + 3 < W) {                                    // all four inside, at a byte address of any alignment: one unaligned 12-byte store
                __builtin_memcpy(dst + o, &d, 12);
            } else {
                const int nb = 3 * min(4, W - x0);                       // W % 4 != 0: byte by byte, up to the row end
#pragma unroll 1
                for (int k = 0; k < nb; ++k) {
                    const uint32_t word = k < 4 ? d.x : k < 8 ? d.y : d.z;
                    dst[o + k] = (uint8_t)(word >> (8 * (k & 3)));
                }
            }
        }
    }
}

template <bool STAGE_OK>
__global__ __launch_bounds__(64) MF_WARP_ATTR void warp_kernel(const FootPlan* __restrict__ plan, const FootRegion* __restrict__ regions,
                                                               WarpGeom g, const uint8_t* __restrict__ frames,
                                                               const double* __restrict__ records, uint8_t* __restrict__ out,
                                                               const float* __restrict__ edges, int n, int W,
                                                               int H, int C, uint32_t border, int32_t* __restrict__ crop, int32_t* __restrict__ clip)
{
    // XCD-aware footprint order.  Workgroups go to the 8 XCDs round-robin by linear id (blockIdx.x first: gridDim.x is a multiple of
    // 8), and each XCD has its own L2: in raster order the four neighbours of a footprint -- whose staged windows overlap its own
    // by 60 % -- would all run on other XCDs and each L2 would fetch the shared rows again.  Workgroup L of frame blockIdx.y
    // therefore takes footprint (L % 8) * per_xcd + L / 8: every XCD sweeps one contiguous eighth of each frame in raster order.
    // (Everything up to the plan is scalar, with host-made constants and 32-bit offsets: the scalar unit is as loaded as the vector
    // unit in this kernel -- profiles/README.md -- and every s_ instruction here is paid by each of the 2.4 M wavefronts of a clip.)
    const uint32_t f = blockIdx.y;
#ifndef MF_NO_ROT
    // The eighth an XCD sweeps ROTATES with the frame: the footprints along the top and bottom frame border are the expensive ones
    // (per-tap path, crop flags: 2.5 x the average), and with a fixed assignment they all land on XCD 0 and XCD 7, which then
    // finish 9 % after the others (-2.1 % kernel time at config 2, -4 % on the all-hot probe).
    const uint32_t t = ((blockIdx.x + f) & 7u) * g.per_xcd + (blockIdx.x >> 3);
#else
    const uint32_t t = (blockIdx.x & 7u) * g.per_xcd + (blockIdx.x >> 3);
#endif
    if (t >= g.per_frame) return;
    footprint_body<STAGE_OK, false>(f, t, plan, regions, g, frames, records, out, edges, n, W, H, C, border, crop, clip);
}

// The crop-boundary scan WITHOUT the pixels (mfs.py:1075-1106 depends on the coordinate maps only, i.e. on the cell table): fills
// crop[f] exactly as warp_kernel does, from the footprints that can set a flag at all.  A wavefront looks at SCAN_GROUP consecutive
// footprints (one lane each reads the footprint's region word: anything the plan certified as DEEP or NOFLAG cannot pass any of
// the four tests) and then runs footprint_body<SCAN> -- ownership and coordinates of the general path, nothing else -- on the
// remaining ones, one after the other: ~4 % of the footprints of a stabilised 1080p clip (the ring along the frame border).
// A host that knows the clip-level rectangle before the first pixel moves can crop + resize each chunk right behind its warp
// (csrc/hostpipe.hip) and, at N > 1, overlap the 16-byte all-reduce with the warp.
constexpr uint32_t SCAN_GROUP = 16;
__global__ __launch_bounds__(64) void crop_scan_kernel(const FootPlan* __restrict__ plan, const FootRegion* __restrict__ regions,
                                                       WarpGeom g, const double* __restrict__ records, const float* __restrict__ edges,
                                                       uint32_t total, int n, int W, int H, int C, int32_t* __restrict__ crop, int32_t* __restrict__ clip)
{
    const uint32_t base = blockIdx.x * SCAN_GROUP, mine = base + threadIdx.x;
    bool need = false;
    if (threadIdx.x < SCAN_GROUP && mine < total) need = (regions[mine].flags_origin & (MF_REGION_DEEP | MF_REGION_NOFLAG)) == 0;
    uint64_t todo = __ballot(need);
    while (todo) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t fp = __builtin_amdgcn_readfirstlane(base + bit);
        const uint32_t f = fp / g.per_frame, t = fp - f * g.per_frame;
        footprint_body<false, true>(f, t, plan, regions, g, nullptr, records, nullptr, edges, n, W, H, C, 0u, crop, clip);
        __builtin_amdgcn_wave_barrier();             // (the next footprint reuses the wavefront's s_hi rows)
    }
}

#ifdef MF_EXP_PHASES
}  // namespace mf
extern "C" int mf_exp_phase_read(unsigned long long out[8], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mf::mf_exp_phase), 8 * sizeof(unsigned long long)) != hipSuccess) return -2;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mf::mf_exp_phase), z, sizeof z) != hipSuccess) return -2; }
    return 0;
}
namespace mf {
#endif

// Self-test of recip_unit_range against IEEE division: counts mismatching bit patterns.
__global__ void selftest_recip_kernel(unsigned long long n, unsigned long long seed, unsigned long long* mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        // mantissa from the hash, exponent -1 or 0 (|w| in [0.5, 2)), random sign; a few exact end points
        unsigned long long bits = (z & 0x800FFFFFFFFFFFFFull) | ((0x3FEull + ((z >> 52) & 1ull)) << 52);
        if ((i & 0xFFFFF) == 0) bits = 0x4000000000000000ull;             // 2.0
        if ((i & 0xFFFFF) == 1) bits = 0x3FE0000000000000ull;             // 0.5
        const double w = __longlong_as_double((long long)bits);
        const double a = recip_unit_range(w);
        const double b = 1.0 / w;
        if (__double_as_longlong(a) != __double_as_longlong(b)) ++bad;
        // the neighbour-pixel route: w_j = w + j h6 with |h6| / w^2 up to the limit the kernel accepts (a second hash draws h6)
        unsigned long long z2 = (z ^ 0xD6E8FEB86659FD93ull) * 0x9E3779B97F4A7C15ull;
        z2 ^= z2 >> 29;
        const double frac = (double)(long long)(z2 >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;        // (-1, 1)
        const double scale = (i & 7) == 0 ? 1.0 : (double)((z2 & 1023) + 1) * (1.0 / 1024.0);                // often small, sometimes the limit
        const double h6 = frac * scale * RECIP_GUESS_LIMIT * (w * w);
        const double c1 = h6 * (a * a), c2 = (h6 * c1) * a;
        if (fabs(c1) <= RECIP_GUESS_LIMIT) {
            const double j = (double)(1 + (int)(z2 % 3));
            const double wj = w + j * h6;
            if (fabs(wj) >= 0.5 && fabs(wj) < 2.0) {
                const double g = recip_from_guess(wj, recip_guess(a, c1, c2, j));
                const double q = 1.0 / wj;
                if (__double_as_longlong(g) != __double_as_longlong(q)) ++bad;
            }
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int launch_selftest_recip(unsigned long long n, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t st)
{
    hipLaunchKernelGGL(selftest_recip_kernel, dim3(2048), dim3(256), 0, st, n, seed, d_mismatches);
    return hip_fail(hipGetLastError(), "selftest_recip_kernel launch");
}

// Self-test of the cheap coordinate chain (coords_fast) against cv2.perspectiveTransform's own arithmetic on hashed matrices
// and positions that satisfy the plan's MF_PLAN_FAST64 / UNIT / DEEP premises: `missed` counts float32 results that differ from
// the exact chain's WITHOUT their midpoint key raising the flag (must be 0), `flagged` the values whose key did (the fallback rate).
// (`margin`, optional: the largest distance between a cheap value and the exact chain's, in float64 ulps of the latter, as double bits)
__global__ void selftest_fast64_kernel(unsigned long long n, unsigned long long seed, unsigned long long* counters, unsigned long long* margin)
{
#ifdef MF_NO_FAST64
    (void)n; (void)seed; (void)counters; (void)margin;          // (A/B build without the cheap chain: nothing to test)
#else
    unsigned long long missed = 0, flagged = 0, tested = 0;
    double far = 0.0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        double r[11];
        for (int q = 0; q < 11; ++q) {                          // eleven uniform draws in (-1, 1)
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            r[q] = (double)(long long)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
            z += 0x9E3779B97F4A7C15ull;
        }
        // near-identity inverse homography with shift, shear and perspective terms up to what the certificates admit
        const double amp = (i & 3) == 0 ? 1.0 : 0.1;             // a quarter of the cases at the limits
        double Hi[9] = { 1.0 + 0.2 * amp * r[0], 0.2 * amp * r[1], 80.0 * r[2], 0.2 * amp * r[3], 1.0 + 0.2 * amp * r[4], 80.0 * r[5],
                         2.0e-4 * amp * r[6], 2.0e-4 * amp * r[7], 1.0 + 0.3 * amp * r[10] };
        const double xs0 = (double)(4 * (int)((r[8] * 0.5 + 0.5) * 2047.0)), yy = (double)(int)((r[9] * 0.5 + 0.5) * 8191.0);
        // the premises, in float64 on the lane's four pixels (the plan checks them on the footprint's corners); half of the cases
        // step along y (the transposed lanes of the pair path)
        const bool vert = ((i >> 2) & 1) != 0;
        const double x0d = vert ? (double)(int)((r[8] * 0.5 + 0.5) * 8191.0) : xs0, y0d = vert ? (double)(4 * (int)((r[9] * 0.5 + 0.5) * 2047.0)) : yy;
        bool ok = true;
        for (int j = 0; j < 4 && ok; ++j) {
            const double x = vert ? x0d : x0d + j, y = vert ? y0d + j : y0d;
            const double w = (x * Hi[6] + y * Hi[7]) + Hi[8];
            const double nx = (x * Hi[0] + y * Hi[1]) + Hi[2], ny = (x * Hi[3] + y * Hi[4]) + Hi[5];
            ok = w > 0.52 && w < 1.9 &&
                 fabs(Hi[0]) * x + fabs(Hi[1]) * y + fabs(Hi[2]) <= 8.0 * nx && fabs(Hi[3]) * x + fabs(Hi[4]) * y + fabs(Hi[5]) <= 8.0 * ny &&
                 fabs(Hi[6]) * x + fabs(Hi[7]) * y + fabs(Hi[8]) <= 2.5 && nx / w >= 1.0 && ny / w >= 1.0 && nx / w < 32768.0 && ny / w < 32768.0;
        }
        if (!ok) continue;
        float u[4], v[4];
        uint32_t keys[8];
        double raw[8];
        if (vert) (void)coords_fast_dir<true>(Hi, x0d, y0d, u, v, keys, raw);
        else (void)coords_fast_dir<false>(Hi, x0d, y0d, u, v, keys, raw);
        for (int j = 0; j < 4; ++j) {
            const double x = vert ? x0d : x0d + j, y = vert ? y0d + j : y0d;
            const double w = (x * Hi[6] + y * Hi[7]) + Hi[8];
            const double iw = 1.0 / w;
            const double ued = ((x * Hi[0] + y * Hi[1]) + Hi[2]) * iw, ved = ((x * Hi[3] + y * Hi[4]) + Hi[5]) * iw;
            const float ue = (float)ued, ve = (float)ved;
            far = fmax(far, fmax(ldexp(fabs(raw[2 * j] - ued), 52 - ilogb(ued)), ldexp(fabs(raw[2 * j + 1] - ved), 52 - ilogb(ved))));
            tested += 2;
            if (keys[2 * j] < FAST64_NEAR) ++flagged; else if (__float_as_uint(ue) != __float_as_uint(u[j])) ++missed;
            if (keys[2 * j + 1] < FAST64_NEAR) ++flagged; else if (__float_as_uint(ve) != __float_as_uint(v[j])) ++missed;
        }
    }
    if (missed) atomicAdd(&counters[0], missed);
    if (flagged) atomicAdd(&counters[1], flagged);
    if (tested) atomicAdd(&counters[2], tested);
    if (margin) atomicMax(margin, (unsigned long long)__double_as_longlong(far));       // (non-negative doubles order like their bits)
#endif
}

int launch_selftest_fast64(unsigned long long n, unsigned long long seed, unsigned long long* d_counters, hipStream_t st, unsigned long long* d_margin)
{
    hipLaunchKernelGGL(selftest_fast64_kernel, dim3(2048), dim3(256), 0, st, n, seed, d_counters, d_margin);
    return hip_fail(hipGetLastError(), "selftest_fast64_kernel launch");
}

// The byte taps (gather_blend_staged here, the staged rows in resize.hip) rely on ds_read_u8_d16_hi ZEROING the low half of its
// destination, which is what a device with SRAM ECC does (every MI300-class part; measured on the MI355X) -- without SRAM ECC the
// low half would be preserved.  Checked once per process on the device in use; a device that behaves differently is refused.
__device__ int d16_probe_result;
__global__ void d16_probe_kernel()
{
    __shared__ uint8_t s[8];
    if (threadIdx.x < 8) s[threadIdx.x] = (uint8_t)(0x11 * (threadIdx.x + 1));
    __syncthreads();
    uint32_t r = 0xFFFFFFFFu;
    const uint32_t at = (uint32_t)(uintptr_t)&s[0];
    asm volatile("ds_read_u8_d16_hi %0, %1 offset:3\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(at) : "memory");
    if (threadIdx.x == 0) d16_probe_result = r == 0x00440000u ? 1 : -1;
}
int check_d16_zero_fill(hipStream_t st)
{
    // per DEVICE: 0 unknown, 1 fine, -1 refused  (benign race: every thread computes the same).  The probe synchronises, so
    // mf_set_device() runs it ahead of time; a launcher only gets here first when the host selected the device itself.
    static std::atomic<int> state[64] = {};           // (per-device host threads may get here together)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hip_fail(hipErrorInvalidDevice, "d16 probe: hipGetDevice");
    if (state[dev] == 0) {
        int h = 0;
        hipLaunchKernelGGL(d16_probe_kernel, dim3(1), dim3(64), 0, st);
        hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpyFromSymbol(&h, HIP_SYMBOL(d16_probe_result), sizeof h);
        if (e != hipSuccess) return hip_fail(e, "d16 probe");
        state[dev] = h == 1 ? 1 : -1;
    }
    if (state[dev] != 1) {
        set_error("device %d preserves the other half of a d16 LDS load (no SRAM ECC): the byte-tap kernels are not built for it", dev);
        return MF_ERR_INVALID_ARG;
    }
    return MF_OK;
}

// Launch constants of warp_kernel / crop_scan_kernel; returns the frames one launch may cover (the grid's y extent, and 32-bit byte
// offsets into the plan -- 16 B per footprint -- and the records), 0 when a single frame is already too large.
static uint32_t make_warp_geom(int W, int H, int R, int C, WarpGeom& g)
{
    const uint32_t nfx = (uint32_t)((W + FOOT_W - 1) / FOOT_W), nfy = (uint32_t)((H + FOOT_H - 1) / FOOT_H);
    const FastDiv by_row = make_fast_div(nfx);
    g.per_frame = nfx * nfy;
    g.per_xcd = (g.per_frame + 7u) / 8u;
    g.nfx = nfx;
    g.div_m = by_row.m; g.div_s = by_row.s; g.div_pass = nfx == 1u ? 0xFFFFFFFFu : 0u;
    g.frame_bytes = 3u * (uint32_t)W * (uint32_t)H;
    g.row_bytes = 3u * (uint32_t)W;
    g.rec_frame_bytes = (uint32_t)(R * C) * (uint32_t)(MF_CELL_DOUBLES * sizeof(double));
    g.edge_frame_bytes = (uint32_t)(R * C) * (uint32_t)(MF_EDGE_FLOATS * sizeof(float));
    // cell column under pixel x of the unwarped grid ~ floor(x C / (W - 1)) = mulhi(x, 2^32 C / (W - 1)) (a guess: the plan decides)
    g.cell_mul_x = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (((uint64_t)C) << 32) / (uint64_t)(W - 1));
    g.cell_mul_y = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (((uint64_t)R) << 32) / (uint64_t)(H - 1));
    g.mesh_cols = (uint32_t)C;
    g.cell_last = (uint32_t)(R * C - 1);
    const uint64_t cap = 0xFFFFFFFFull;
    uint64_t per_launch = 65535;
    per_launch = per_launch < cap / (16ull * g.per_frame) ? per_launch : cap / (16ull * g.per_frame);
    per_launch = per_launch < cap / g.rec_frame_bytes ? per_launch : cap / g.rec_frame_bytes;
    return (uint32_t)per_launch;
}

int launch_warp(const uint8_t* frames, uint8_t* out, const TableView& tv, int n, int W, int H, int R, int C,
                uint32_t border, int32_t* crop, hipStream_t st)
{
    if (const int rc = check_d16_zero_fill(st)) return rc;
    // (any number of frames: the launches below take at most 65,535 -- the grid's y extent -- at a time)
    if (n <= 0 || W < 2 || H < 2 || W > 32767 || H > 32767 || R <= 0 || C <= 0 || R > MAX_MESH ||
        C > MAX_MESH) {
        set_error("mf_warp_u8c3: unsupported shape n=%d W=%d H=%d R=%d C=%d", n, W, H, R, C);
        return MF_ERR_INVALID_ARG;
    }
    WarpGeom g;
    uint64_t per_launch = make_warp_geom(W, H, R, C, g);
    if (per_launch == 0) {
        set_error("mf_warp_u8c3: frame too large");
        return MF_ERR_INVALID_ARG;
    }
    if (const char* e = getenv("MF_WARP_FRAMES_PER_LAUNCH")) {      // testing aid: forces the multi-launch split on small clips
        const long v = atol(e);
        if (v > 0 && (uint64_t)v < per_launch) per_launch = (uint64_t)v;
    }
    // staging reads dword-aligned 16-byte chunks: needs a 4-byte aligned clip (W % 4 == 0 is checked by the plan)
    const bool stage_ok = ((uintptr_t)frames & 3u) == 0;
    for (int f0 = 0; f0 < n; f0 += (int)per_launch) {
        const int m = n - f0 < (int)per_launch ? n - f0 : (int)per_launch;
        const dim3 grid(g.per_xcd * 8u, (uint32_t)m);                  // one wavefront per 32 x 8 footprint
        const uint8_t* fr = frames + (size_t)f0 * g.frame_bytes;
        uint8_t* o = out + (size_t)f0 * g.frame_bytes;
        const double* rec = tv.records + (size_t)f0 * R * C * MF_CELL_DOUBLES;
        const float* ed = tv.edges + (size_t)f0 * R * C * MF_EDGE_FLOATS;
        const FootPlan* pl = tv.plan + (size_t)f0 * g.per_frame;
        const FootRegion* rgn = tv.regions + (size_t)f0 * g.per_frame;
        if (stage_ok)
            hipLaunchKernelGGL(warp_kernel<true>, grid, dim3(64), 0, st, pl, rgn, g, fr, rec, o, ed, m, W, H, C, border, crop + 4 * (size_t)f0, tv.bounds);
        else
            hipLaunchKernelGGL(warp_kernel<false>, grid, dim3(64), 0, st, pl, rgn, g, fr, rec, o, ed, m, W, H, C, border, crop + 4 * (size_t)f0, tv.bounds);
    }
    return hip_fail(hipGetLastError(), "warp_kernel launch");
}

int launch_crop_scan(const TableView& tv, int n, int W, int H, int R, int C, int32_t* crop, hipStream_t st)
{
    // (any number of frames, like launch_warp: the launches below take per_launch frames at a time)
    if (n <= 0 || W < 2 || H < 2 || W > 32767 || H > 32767 || R <= 0 || C <= 0 || R > MAX_MESH || C > MAX_MESH) {
        set_error("mf_crop_scan_f64: unsupported shape n=%d W=%d H=%d R=%d C=%d", n, W, H, R, C);
        return MF_ERR_INVALID_ARG;
    }
    WarpGeom g;
    const uint32_t per_launch = make_warp_geom(W, H, R, C, g);
    if (per_launch == 0) {
        set_error("mf_crop_scan_f64: frame too large");
        return MF_ERR_INVALID_ARG;
    }
    for (int f0 = 0; f0 < n; f0 += (int)per_launch) {
        const int m = n - f0 < (int)per_launch ? n - f0 : (int)per_launch;
        const uint32_t total = (uint32_t)m * g.per_frame;                  // (per_launch keeps 16 B x total below 2^32)
        hipLaunchKernelGGL(crop_scan_kernel, dim3((total + SCAN_GROUP - 1) / SCAN_GROUP), dim3(64), 0, st,
                           tv.plan + (size_t)f0 * g.per_frame, tv.regions + (size_t)f0 * g.per_frame, g,
                           tv.records + (size_t)f0 * R * C * MF_CELL_DOUBLES, tv.edges + (size_t)f0 * R * C * MF_EDGE_FLOATS, total, m, W, H, C,
                           crop + 4 * (size_t)f0, tv.bounds);
    }
    return hip_fail(hipGetLastError(), "crop_scan_kernel launch");
}

// Clip-level bounds, mfs.py:1103-1106.
__global__ __launch_bounds__(256) void crop_reduce_kernel(const int32_t* __restrict__ crop, int n, int W, int H,
                                                          int32_t* __restrict__ bounds)
{
    __shared__ int32_t red[4][256];
    int l = 0, t = 0, r = W - 1, b = H - 1;
    for (int i = threadIdx.x; i < n; i += 256) {
        l = max(l, crop[4 * i + 0]); t = max(t, crop[4 * i + 1]);
        r = min(r, crop[4 * i + 2]); b = min(b, crop[4 * i + 3]);
    }
    red[0][threadIdx.x] = l; red[1][threadIdx.x] = t; red[2][threadIdx.x] = r; red[3][threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = max(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = min(red[3][threadIdx.x], red[3][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) bounds[threadIdx.x] = red[threadIdx.x][0];
}

int launch_crop_reduce(const int32_t* crop, int n, int W, int H, int32_t* bounds, hipStream_t st)
{
    if (n <= 0) { set_error("mf_crop_reduce: n=%d", n); return MF_ERR_INVALID_ARG; }
    hipLaunchKernelGGL(crop_reduce_kernel, dim3(1), dim3(256), 0, st, crop, n, W, H, bounds);
    return hip_fail(hipGetLastError(), "crop_reduce_kernel launch");
}

}  // namespace mf


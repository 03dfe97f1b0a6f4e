// Vertex-motion accumulation: matched features + one global homography per frame pair -> per-vertex
// velocities (float32) and their running sum (float64), i.e. the tensor the Jacobi sweep starts from.
//
// Reference (meshflowstabilizer.py): feature residual velocities :420; ellipse around each feature :426-448;
// per-vertex statistics.median of the covering residuals, 0 when none :338-353; global vertex motion
// (float32 perspectiveTransform of the grid - grid) :324-328; sum -> float32 :354-355; 3x3 cv2.medianBlur
// :359-360; displacement[t+1] = displacement[t] + velocity[t] :271,281.  Feature arrays are float64 in the
// reference's flow (the sub-frame offset added at :578 promotes them), so every step up to :353 is float64.
// CPU oracle: oracle/motion_oracle.py (pinned against the reference) and oracle/motion_oracle.c.
//
// Passes, everything exact (one IEEE rounding per operation, -ffp-contract=off, no reordering):
//   feature_prep_kernel    one thread per feature: residual velocity (as order-preserving 64-bit keys) and, per
//                          mesh row, the column range its ellipse covers
//   bitonic_*_kernel       per frame pair, the features' x keys and y keys sorted once (LDS bitonic network on
//                          4096-element tiles; global compare-exchange steps only for pairs with more features)
//   vertex_row_median_kernel  one workgroup per (pair, mesh row): per-chunk counts of covering features per column from a difference
//                          array, then the median chunk of every column read once more -- statistics.median of every vertex of the
//                          row with ONE gather per feature and order (the per-vertex kernel below remains for oversized inputs)
//   vertex_median_kernel   one wavefront per (pair, vertex): counts the covering features, then walks the pair's
//                          sorted order until the middle covering feature(s) -- statistics.median without building
//                          or sorting a per-vertex list -- and adds the vertex's global motion
//   median_blur_kernel     3x3 median of the raw velocities (replicated borders), one thread per (pair, vertex)
//   accumulate_kernel      running sum over the pairs, one thread per (vertex, component): the additions must
//                          stay sequential in time to round like the reference
#include <stdlib.h>

#include "mf_common.h"

namespace mf {

namespace {

constexpr int kTile = 4096;          // elements one workgroup sorts in LDS (48 KB)
constexpr int kSortThreads = 512;

struct alignas(4) Span { int16_t first, last; };   // columns first..last of one mesh row lie inside a feature's ellipse

// Order-preserving key: unsigned comparison of keys orders the doubles (-0.0 just below +0.0, which never changes
// a median's value); the all-ones key pads the sort.
__device__ __forceinline__ unsigned long long key_of(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? (k ^ 0x8000000000000000ull) : ~k));
}

// One thread per slot of the padded per-pair arrays (npad slots, a power of two >= the pair's feature count).
// skey/sidx: [2P][npad] (segment 2p: x residuals, 2p+1: y residuals); spans: [R+1][total_features].
__global__ __launch_bounds__(256) void feature_prep_kernel(const double* __restrict__ early, const double* __restrict__ late,
                                                           const int32_t* __restrict__ offsets,
                                                           const double* __restrict__ hom, int W, int H, int R, int C,
                                                           double half_rows, double ell_rows, double ell_cols,
                                                           size_t total_features, int npad,
                                                           unsigned long long* __restrict__ skey, uint32_t* __restrict__ sidx,
                                                           Span* __restrict__ spans, int32_t* __restrict__ status)
{
    const int p = blockIdx.y;
    const int k0 = offsets[p], K = offsets[p + 1] - k0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    const size_t sx = (size_t)(2 * p) * npad + i, sy = sx + npad;
    sidx[sx] = (uint32_t)i;
    sidx[sy] = (uint32_t)i;
    if (i >= K) {
        skey[sx] = ~0ull;
        skey[sy] = ~0ull;
        return;
    }
    const double* __restrict__ m = hom + 9 * (size_t)p;
    const size_t g = (size_t)k0 + i;
    const double x = early[2 * g], y = early[2 * g + 1];
    // cv2.perspectiveTransform, CV_64F (mfs.py:420)
    const double w = (x * m[6] + y * m[7]) + m[8];
    double tx = 0.0, ty = 0.0;
    if (fabs(w) > 1.1920928955078125e-07) {
        const double iw = 1.0 / w;
        tx = ((x * m[0] + y * m[1]) + m[2]) * iw;
        ty = ((x * m[3] + y * m[4]) + m[5]) * iw;
    }
    skey[sx] = key_of(late[2 * g] - tx);
    skey[sy] = key_of(late[2 * g + 1] - ty);

    const double frow = (y / (double)H) * (double)R;          // mfs.py:426
    const double fcol = (x / (double)W) * (double)C;          // mfs.py:427
    const double lo = frow - half_rows, hi = frow + half_rows;
    for (int r = 0; r <= R; ++r) {
        const double rd = (double)r;
        Span sp;
        sp.first = 1;
        sp.last = 0;
        // ceil(lo) <= r <= floor(hi)  <=>  lo <= r <= hi   (r is an integer; mfs.py:438-441)
        if (lo <= rd && rd <= hi) {
            const double q = (rd - frow) / ell_rows;
            const double s = 0.25 - q * q;
            if (s < 0.0) {
                atomicOr(status, 1);                          // math.sqrt raises ValueError here (mfs.py:444)
            } else {
                const double hw = ell_cols * sqrt(s);
                const double a = fcol - hw, b = fcol + hw;
                // max(0, ceil(a)) .. min(C, floor(b))   (mfs.py:445-446)
                sp.first = (int16_t)(a <= 0.0 ? 0 : (a > (double)C ? C + 1 : (int)ceil(a)));
                sp.last = (int16_t)(b >= (double)C ? C : (b < 0.0 ? -1 : (int)floor(b)));
            }
        }
        spans[(size_t)r * total_features + g] = sp;
    }
}

// Bitonic network on one tile of a segment, in LDS: for k = k_lo .. k_hi (doubling) the compare-exchange steps with
// distance j = min(k/2, tile/2) .. 1.  Direction of a pair = bit k of its position in the segment (ascending when 0).
__global__ __launch_bounds__(kSortThreads) void bitonic_tile_kernel(unsigned long long* __restrict__ skey, uint32_t* __restrict__ sidx,
                                                                   int npad, int k_lo, int k_hi)
{
    __shared__ unsigned long long s_key[kTile];
    __shared__ uint32_t s_idx[kTile];
    const int len = npad < kTile ? npad : kTile;
    const int base = blockIdx.x * kTile;
    unsigned long long* gk = skey + (size_t)blockIdx.y * npad + base;
    uint32_t* gi = sidx + (size_t)blockIdx.y * npad + base;
    for (int t = threadIdx.x; t < len; t += kSortThreads) {
        s_key[t] = gk[t];
        s_idx[t] = gi[t];
    }
    __syncthreads();
    for (int k = k_lo; k <= k_hi; k <<= 1) {
        for (int j = (k >> 1) < (len >> 1) ? (k >> 1) : (len >> 1); j >= 1; j >>= 1) {
            for (int t = threadIdx.x; t < (len >> 1); t += kSortThreads) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo + j;
                const bool up = ((base + lo) & k) == 0;
                const unsigned long long a = s_key[lo], b = s_key[hi];
                if ((a > b) == up) {
                    s_key[lo] = b;
                    s_key[hi] = a;
                    const uint32_t ia = s_idx[lo];
                    s_idx[lo] = s_idx[hi];
                    s_idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < len; t += kSortThreads) {
        gk[t] = s_key[t];
        gi[t] = s_idx[t];
    }
}

// One compare-exchange step with distance j >= kTile of the merge with block size k, in global memory.
__global__ __launch_bounds__(256) void bitonic_global_kernel(unsigned long long* __restrict__ skey, uint32_t* __restrict__ sidx,
                                                             int npad, int j, int k)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (npad >> 1)) return;
    unsigned long long* gk = skey + (size_t)blockIdx.y * npad;
    uint32_t* gi = sidx + (size_t)blockIdx.y * npad;
    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo + j;
    const bool up = (lo & k) == 0;
    const unsigned long long a = gk[lo], b = gk[hi];
    if ((a > b) == up) {
        gk[lo] = b;
        gk[hi] = a;
        const uint32_t ia = gi[lo];
        gi[lo] = gi[hi];
        gi[hi] = ia;
    }
}

// Position (in the sorted order of one axis) of the k1-th and k2-th covering feature of vertex column c.
// Every lane returns the same two keys.
__device__ __forceinline__ void walk_sorted(const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                            const Span* __restrict__ row_spans, int K, int c, int k1, int k2, int lane,
                                            unsigned long long& key1, unsigned long long& key2)
{
    int count = 0;
    bool have1 = false;
    key1 = key2 = 0;
    for (int base = 0; base < K; base += 64) {
        const int j = base + lane;
        bool cover = false;
        if (j < K) {
            const Span sp = row_spans[sidx[j]];
            cover = sp.first <= c && c <= sp.last;
        }
        const unsigned long long m = __ballot(cover);
        const int cnt = __popcll(m);
        if (count + cnt > k1) {
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            const unsigned long long mine = j < K ? skey[j] : 0ull;
            if (!have1) {
                const unsigned long long hit = __ballot(cover && count + before == k1);
                key1 = __shfl(mine, __ffsll((long long)hit) - 1);
                have1 = true;
            }
            if (count + cnt > k2) {
                const unsigned long long hit = __ballot(cover && count + before == k2);
                key2 = __shfl(mine, __ffsll((long long)hit) - 1);
                return;
            }
        }
        count += cnt;
    }
}

// One wavefront per (pair, vertex); 4 per workgroup.
__global__ __launch_bounds__(256) void vertex_median_kernel(const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                                            const Span* __restrict__ spans, const int32_t* __restrict__ offsets,
                                                            const double* __restrict__ hom, int P, int W, int H, int R, int C,
                                                            size_t total_features, int npad, float2* __restrict__ raw)
{
    const int lane = threadIdx.x & 63;
    const int C1 = C + 1, V = (R + 1) * C1;
    const int task = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (task >= P * V) return;
    const int p = task / V, v = task - p * V;
    const int r = v / C1, c = v - r * C1;
    const int k0 = offsets[p], K = offsets[p + 1] - k0;
    const Span* __restrict__ row_spans = spans + (size_t)r * total_features + k0;

    // (1) how many features cover this vertex (mfs.py:438-450)
    int n = 0;
    for (int base = 0; base < K; base += 64) {
        const int i = base + lane;
        bool cover = false;
        if (i < K) {
            const Span sp = row_spans[i];
            cover = sp.first <= c && c <= sp.last;
        }
        n += __popcll(__ballot(cover));
    }

    // (2) statistics.median of their residuals (mfs.py:338-353): covering features number (n-1)/2 and n/2 in sorted order
    double medx = 0.0, medy = 0.0;
    if (n > 0) {
        const int k1 = (n - 1) >> 1, k2 = n >> 1;
        unsigned long long x1, x2, y1, y2;
        walk_sorted(skey + (size_t)(2 * p) * npad, sidx + (size_t)(2 * p) * npad, row_spans, K, c, k1, k2, lane, x1, x2);
        walk_sorted(skey + (size_t)(2 * p + 1) * npad, sidx + (size_t)(2 * p + 1) * npad, row_spans, K, c, k1, k2, lane, y1, y2);
        medx = (n & 1) ? value_of(x1) : (value_of(x1) + value_of(x2)) / 2.0;
        medy = (n & 1) ? value_of(y1) : (value_of(y1) + value_of(y2)) / 2.0;
    }

    // (3) + the vertex's global motion (mfs.py:324-328, 354-355)
    if (lane == 0) {
        const double* __restrict__ m = hom + 9 * (size_t)p;
        const float gxf = (float)ceil((double)(W - 1) * ((double)c / (double)C));
        const float gyf = (float)ceil((double)(H - 1) * ((double)r / (double)R));
        const double gx = (double)gxf, gy = (double)gyf;
        const double w = (gx * m[6] + gy * m[7]) + m[8];
        float px = 0.0f, py = 0.0f;
        if (fabs(w) > 1.1920928955078125e-07) {
            const double iw = 1.0 / w;
            px = (float)(((gx * m[0] + gy * m[1]) + m[2]) * iw);
            py = (float)(((gx * m[3] + gy * m[4]) + m[5]) * iw);
        }
        const float globx = px - gxf, globy = py - gyf;
        raw[task] = make_float2((float)((double)globx + medx), (float)((double)globy + medy));
    }
}

// The same result for a whole MESH ROW per wavefront (pair p, row r: its C + 1 vertices share the row's spans and the pair's two sorted
// orders), reading every span ONCE per order instead of once per vertex and order:
//   pass over the sorted-x order, 64 features at a time: every lane adds its feature's column interval to a difference array in LDS
//     (two ds_add: +1 at `first`, -1 behind `last`), a wave prefix sum turns it into the chunk's count per column, and the running
//     totals BEFORE each chunk go to a table cum[chunk][column]; the same over the sorted-y order;
//   per column: the chunk that holds covering feature number (n-1)/2 (and n/2) is found in the table, and only that chunk is read
//     again to pick the key -- statistics.median (mfs.py:338-353) of exactly the same features in exactly the same order as the
//     per-vertex kernel above.
// 17 x fewer span gathers at a 16 x 16 mesh.  For up to 64 chunks (4096 features per pair) and 64 columns; the per-vertex kernel
// takes everything else.
constexpr int kRowChunks = 64;
// (workgroup = two wavefronts: wavefront 0 works through the sorted-x order, wavefront 1 through the sorted-y order, both the counting
// pass and the per-column picks; they meet once, before the output)
__global__ __launch_bounds__(128) void vertex_row_median_kernel(const unsigned long long* __restrict__ skey, const uint32_t* __restrict__ sidx,
                                                                const Span* __restrict__ spans, const int32_t* __restrict__ offsets,
                                                                const double* __restrict__ hom, int W, int H, int R, int C,
                                                                size_t total_features, int npad, float2* __restrict__ raw)
{
    extern __shared__ int s_row[];                           // cum[2][nch + 1][C1] | diff[2][C1 + 1] | med[2][C1] (double)
    const int lane = threadIdx.x & 63;
    const int ax = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C1 = C + 1;
    const int p = blockIdx.x / (R + 1), r = blockIdx.x - p * (R + 1);
    const int k0 = offsets[p], K = offsets[p + 1] - k0;
    const int nch = (K + 63) >> 6;
    const Span* __restrict__ row_spans = spans + (size_t)r * total_features + k0;
    int* table = s_row + ax * (nch + 1) * C1;
    int* diff = s_row + 2 * (nch + 1) * C1 + ax * (C1 + 1);
    double* med = reinterpret_cast<double*>(s_row + ((2 * (nch + 1) * C1 + 2 * (C1 + 1) + 1) & ~1));
    const unsigned long long* __restrict__ keys = skey + (size_t)(2 * p + ax) * npad;
    const uint32_t* __restrict__ order = sidx + (size_t)(2 * p + ax) * npad;

    int running = 0;                                         // lane c: covering features of column c so far
    for (int i = 0; i < nch; ++i) {
        if (lane < C1) table[i * C1 + lane] = running;
        if (lane <= C1) diff[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (this wavefront's own LDS region: its LDS operations execute in order)
        const int j = (i << 6) + lane;
        if (j < K) {
            const Span sp = row_spans[order[j]];
            if (sp.first <= sp.last) {                       // (0 <= first <= last <= C for a non-empty span)
                atomicAdd(&diff[sp.first], 1);
                atomicAdd(&diff[sp.last + 1], -1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (this wavefront's own LDS region: its LDS operations execute in order)
        int v = lane < C1 ? diff[lane] : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(v, off);
            if (lane >= off) v += t;
        }
        running += v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (this wavefront's own LDS region: its LDS operations execute in order)
    }
    if (lane < C1) table[nch * C1 + lane] = running;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    double mine_med = 0.0;                                   // lane c keeps column c's median on this wavefront's axis
    for (int c = 0; c < C1; ++c) {
        const int n = table[nch * C1 + c];
        if (n == 0) continue;                                 // (wave-uniform)
        const int k1 = (n - 1) >> 1, k2 = n >> 1;
        unsigned long long key[2];
        for (int q = 0; q < 2; ++q) {
            const int k = q ? k2 : k1;
            if (q && k2 == k1) { key[1] = key[0]; continue; }
            // the chunk with cum[i] <= k < cum[i + 1]
            const bool here = lane < nch && table[lane * C1 + c] <= k && k < table[(lane + 1) * C1 + c];
            const int i = __ffsll((long long)__ballot(here)) - 1;
            const int before_chunk = table[i * C1 + c];
            const int j = (i << 6) + lane;
            bool cover = false;
            unsigned long long kj = 0ull;
            if (j < K) {
                const Span sp = row_spans[order[j]];
                cover = sp.first <= c && c <= sp.last;
                kj = keys[j];
            }
            const unsigned long long m = __ballot(cover);
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            const unsigned long long hit = __ballot(cover && before_chunk + before == k);
            key[q] = __shfl(kj, __ffsll((long long)hit) - 1);
        }
        const double mv = (n & 1) ? value_of(key[0]) : (value_of(key[0]) + value_of(key[1])) / 2.0;
        if (lane == c) mine_med = mv;
    }
    if (lane < C1) med[ax * C1 + lane] = mine_med;
    __syncthreads();

    // + the vertex's global motion (mfs.py:324-328, 354-355): lane c of wavefront 0 writes vertex (r, c)
    if (ax == 0 && lane < C1) {
        const int c = lane;
        const double medx = med[c], medy = med[C1 + c];
        const double* __restrict__ m = hom + 9 * (size_t)p;
        const float gxf = (float)ceil((double)(W - 1) * ((double)c / (double)C));
        const float gyf = (float)ceil((double)(H - 1) * ((double)r / (double)R));
        const double gx = (double)gxf, gy = (double)gyf;
        const double w = (gx * m[6] + gy * m[7]) + m[8];
        float px = 0.0f, py = 0.0f;
        if (fabs(w) > 1.1920928955078125e-07) {
            const double iw = 1.0 / w;
            px = (float)(((gx * m[0] + gy * m[1]) + m[2]) * iw);
            py = (float)(((gx * m[3] + gy * m[4]) + m[5]) * iw);
        }
        const float globx = px - gxf, globy = py - gyf;
        raw[(size_t)p * (R + 1) * C1 + (size_t)r * C1 + c] = make_float2((float)((double)globx + medx), (float)((double)globy + medy));
    }
}

__device__ __forceinline__ void order(float& a, float& b)
{
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}

__device__ __forceinline__ float median9(float (&w)[9])
{
    // bubble network: after pass i the i largest values are in place; 5 passes put the median at w[4]
#pragma unroll
    for (int pass = 0; pass < 5; ++pass)
#pragma unroll
        for (int j = 0; j + 1 < 9 - pass; ++j) order(w[j], w[j + 1]);
    return w[4];
}

__global__ __launch_bounds__(256) void median_blur_kernel(const float2* __restrict__ raw, float2* __restrict__ vel, int P, int R, int C)
{
    const int C1 = C + 1, V = (R + 1) * C1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P * V) return;
    const int p = t / V, v = t - p * V;
    const int r = v / C1, c = v - r * C1;
    float wx[9], wy[9];
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const int rr = min(max(r + dr, 0), R), cc = min(max(c + dc, 0), C);      // cv2.medianBlur replicates borders
            const float2 s = raw[(size_t)p * V + rr * C1 + cc];
            wx[(dr + 1) * 3 + dc + 1] = s.x;
            wy[(dr + 1) * 3 + dc + 1] = s.y;
        }
    vel[t] = make_float2(median9(wx), median9(wy));
}

__global__ __launch_bounds__(256) void accumulate_kernel(const float* __restrict__ vel, double* __restrict__ disp, int P, int V2)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= V2) return;
    double acc = 0.0;
    disp[s] = 0.0;                                                                   // mfs.py:271
#pragma unroll 8
    for (int p = 0; p < P; ++p) {
        acc = acc + (double)vel[(size_t)p * V2 + s];                                 // mfs.py:281
        disp[(size_t)(p + 1) * V2 + s] = acc;
    }
}

__global__ void selftest_sqrt_kernel(unsigned long long n, unsigned long long seed, unsigned long long* mismatches)
{
    // sqrt(s), s in (0, 0.25], must be the correctly rounded root r: (r - h_dn)^2 < s <= (r + h_up)^2 with h the half
    // distances to r's neighbours.  (r +- h)^2 - s = e +- 2 r h + h^2 with e = r*r - s exact (fma); e and 2 r h are
    // multiples of ulp(r)^2 = 4 h^2, so the sign of e +- 2 r h decides.
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        const double s = 0.25 * ((double)(z >> 11) + 1.0) * (1.0 / 9007199254740992.0);
        const double r = sqrt(s);
        const double e = __builtin_fma(r, r, -s);
        const double up = __longlong_as_double(__double_as_longlong(r) + 1), dn = __longlong_as_double(__double_as_longlong(r) - 1);
        const double hi = e + 2.0 * r * ((up - r) * 0.5);
        const double lo = e - 2.0 * r * ((r - dn) * 0.5);
        if (hi < 0.0 || lo >= 0.0) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

// Workspace: skey [2P][npad] | sidx [2P][npad] | spans [R+1][total] | raw velocities [P][V].
struct MotionWork { unsigned long long* skey; uint32_t* sidx; Span* spans; float2* raw; size_t bytes; int npad; };

static MotionWork motion_work(void* base, int total_features, int max_per_pair, int P, int R, int C)
{
    const size_t total = total_features > 0 ? total_features : 1, V = (size_t)(R + 1) * (C + 1), segs = 2 * (size_t)(P > 0 ? P : 1);
    MotionWork w;
    w.npad = 64;
    while (w.npad < max_per_pair) w.npad <<= 1;
    char* p = (char*)base;
    w.skey = (unsigned long long*)p;
    p += align16(segs * w.npad * sizeof(unsigned long long));
    w.sidx = (uint32_t*)p;
    p += align16(segs * w.npad * sizeof(uint32_t));
    w.spans = (Span*)p;
    p += align16(total * (size_t)(R + 1) * sizeof(Span));
    w.raw = (float2*)p;
    p += align16((size_t)(P > 0 ? P : 1) * V * sizeof(float2));
    w.bytes = (size_t)(p - (char*)base);
    return w;
}

size_t vertex_motion_workspace_bytes(int total_features, int max_per_pair, int P, int R, int C)
{
    return motion_work(nullptr, total_features, max_per_pair, P, R, C).bytes;
}

int launch_vertex_motion(const double* early, const double* late, const int32_t* offsets, const double* hom, int P,
                         int total_features, int max_per_pair, int W, int H, int R, int C, int ell_rows, int ell_cols,
                         float* vel, double* disp, void* work, int32_t* status, hipStream_t st)
{
    if (P < 0 || R <= 0 || C <= 0 || R > 32000 || C > 32000 || W <= 0 || H <= 0 || ell_rows <= 0 || ell_cols <= 0 ||
        total_features < 0 || max_per_pair < 0 || max_per_pair > (1 << 28)) {
        set_error("vertex_motion: bad sizes (P=%d R=%d C=%d W=%d H=%d ellipse=%dx%d)", P, R, C, W, H, ell_rows, ell_cols);
        return MF_ERR_INVALID_ARG;
    }
    const int V = (R + 1) * (C + 1), V2 = 2 * V;
    const MotionWork w = motion_work(work, total_features, max_per_pair, P, R, C);
    if (P > 0) {
        const int npad = w.npad;
        feature_prep_kernel<<<dim3((npad + 255) / 256, P), 256, 0, st>>>(
            early, late, offsets, hom, W, H, R, C, (double)ell_rows / 2.0, (double)ell_rows, (double)ell_cols,
            (size_t)total_features, npad, w.skey, w.sidx, w.spans, status);
        MF_HIP_TRY(hipGetLastError());
        if (max_per_pair > 1) {
            const int tiles = npad > kTile ? npad / kTile : 1;
            bitonic_tile_kernel<<<dim3(tiles, 2 * P), kSortThreads, 0, st>>>(w.skey, w.sidx, npad, 2, npad < kTile ? npad : kTile);
            MF_HIP_TRY(hipGetLastError());
            for (int k = 2 * kTile; k <= npad; k <<= 1) {
                for (int j = k >> 1; j >= kTile; j >>= 1) {
                    bitonic_global_kernel<<<dim3((npad / 2 + 255) / 256, 2 * P), 256, 0, st>>>(w.skey, w.sidx, npad, j, k);
                    MF_HIP_TRY(hipGetLastError());
                }
                bitonic_tile_kernel<<<dim3(tiles, 2 * P), kSortThreads, 0, st>>>(w.skey, w.sidx, npad, k, k);
                MF_HIP_TRY(hipGetLastError());
            }
        }
        static const bool per_vertex = [] { const char* v = getenv("MF_MEDIAN_PER_VERTEX"); return v && *v == '1'; }();   // testing aid
        const int nch_max = (max_per_pair + 63) / 64;
        if (!per_vertex && nch_max <= kRowChunks && C + 1 <= 64) {
            const size_t lds = ((size_t)2 * (nch_max + 1) * (C + 1) + 2 * (C + 2) + 2) * sizeof(int) + (size_t)2 * (C + 1) * sizeof(double);
            vertex_row_median_kernel<<<P * (R + 1), 128, lds, st>>>(w.skey, w.sidx, w.spans, offsets, hom, W, H, R, C,
                                                                   (size_t)total_features, npad, w.raw);
        } else {
            vertex_median_kernel<<<(P * V + 3) / 4, 256, 0, st>>>(w.skey, w.sidx, w.spans, offsets, hom, P, W, H, R, C,
                                                                (size_t)total_features, npad, w.raw);
        }
        MF_HIP_TRY(hipGetLastError());
        median_blur_kernel<<<(P * V + 255) / 256, 256, 0, st>>>(w.raw, (float2*)vel, P, R, C);
        MF_HIP_TRY(hipGetLastError());
    }
    accumulate_kernel<<<(V2 + 255) / 256, 256, 0, st>>>(vel, disp, P, V2);
    MF_HIP_TRY(hipGetLastError());
    return MF_OK;
}

int launch_selftest_sqrt(unsigned long long n, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t st)
{
    selftest_sqrt_kernel<<<1024, 256, 0, st>>>(n, seed, d_mismatches);
    MF_HIP_TRY(hipGetLastError());
    return MF_OK;
}

}  // namespace mf

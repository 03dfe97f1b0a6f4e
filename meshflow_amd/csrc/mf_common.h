// Shared declarations of libmeshflow_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/meshflow_hip.h"

namespace mf {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define MF_HIP_TRY(expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return mf::hip_fail(_e, #expr); \
    } while (0)

// Compact per-cell bounding box that follows the records in the cell table.
struct alignas(8) CellBox { int16_t x0, y0, x1, y1; };

inline size_t table_records(int n, int R, int C) { return (size_t)n * R * C; }
// wavefronts (64 cells each) the cell-table kernel spends on one frame = reach slots per frame
__host__ __device__ inline int reach_parts(int R, int C) { return (R * C + 63) / 64; }

// Layout of the cell table blob (mf_cell_table_bytes): records | boxes | edges | uedges | plan | regions | reach | grid.
//   records: n*R*C x MF_CELL_DOUBLES float64      (ABI, include/meshflow_hip.h)
//   boxes:   n*R*C x CellBox                      compact copy of the record's bbox
//   edges:   n*R*C x MF_EDGE_FLOATS float32       4 affine edge functions {a, b, c} scaled by their error bound, then the 4 bounds
//   uedges:  n*R*C x MF_UEDGE_FLOATS float32      the same 4 functions in units of 1/32 pixel (what the plan kernel classifies with)
//   plan:    n x ceil(H/8) x ceil(W/32) x 16 B    per 32x8-pixel footprint: candidate cells, descending
//   regions: n x ceil(H/8) x ceil(W/32) x 8 B     per footprint: source region the warp kernel stages in LDS
//   reach:   n x reach_parts(R, C) x 4 int32      per frame and cell-table wavefront: max extent of a box beyond its grid rect
//   grid:    (C+1) + (R+1) int32                  vertex x / y pixel coordinates
//   bounds:  4 int32                              clip-level rectangle {left, top, right, bottom} of the table's frames: set to the defaults
//                                                 by the cell-table kernel, folded into by every warp / crop-scan launch on the table
#define MF_EDGE_FLOATS 16
#define MF_UEDGE_FLOATS 12
#define MF_FOOT_W 32
#define MF_FOOT_H 8
// Plan entry (uint16): bits 0-11 cell index, bit 14 = entry valid, bit 15 = IN (every pixel of the footprint
// passes the cell's mask test).  Up to 8 entries in descending cell order; the list ends at the first IN
// entry or the first invalid one.  Entry 7 == 0xFFFF: too many candidates -- entries 0-3 then hold the
// cell range r_lo, r_hi, c_lo, c_hi and the warp kernel tests every cell of the range.
// Lists of at most 4 entries carry an edge code per entry in e[4 + i] (bit 15 set, bit 14 clear), bits 0-5: 0-3 = of the
// cell's four mask-edge functions only this one can fail inside the footprint (the others are > +1 at all four
// corners); 4 = test all four; 8 | a | b << 4 = only edges a and b can fail (a < b).
#define MF_PLAN_CODES 0x8000u
#define MF_PLAN_VALID 0x4000u
#define MF_PLAN_IN 0x8000u
#define MF_PLAN_OVERFLOW 0xFFFFu
// A list that consists of ONE IN entry leaves entry 1 unused (bit 14 clear); its bit 0 then certifies the cell's projective
// denominator on this footprint: 0.52 < w < 1.9 at the four corners (hence on every pixel: w is affine) and |h6| <= 0.9 * 2.5e-4 *
// min(w)^2, the condition of the warp kernel's reciprocal guess -- the kernel skips both per-pixel tests.
#define MF_PLAN_UNIT 0x0001u
// ... and MF_PLAN_HOT (only with UNIT) that the footprint needs nothing else either: it lies completely inside the frame and its source
// region is STAGED and DEEP (below).  The warp kernel then runs its straight-line path: no lane masks, no per-pixel checks.
// (bit 13: clear in every valid entry -- cell indices take bits 0-11 -- and in the raw row / column numbers of an overflow list, so
// the kernel tests this one bit without looking at the rest of the list)
#define MF_PLAN_HOT 0x2000u
// ... and MF_PLAN_FAST64 (only with HOT) the premises of the warp kernel's cheap float64 coordinate chain (warp.hip,
// cell_coords_fast): at every pixel of the footprint |h0| x + |h1| y + |h2| <= 8 (h0 x + h1 y + h2), the same for the second row,
// and |h6| x + |h7| y + |h8| <= 2.5.
#define MF_PLAN_FAST64 0x0002u
// ... and MF_PLAN_BORDER in the same unused entry 1 of a ONE-entry list (with or without UNIT / IN; never with HOT): the "border" path of the
// warp kernel.  The single candidate is IN or MIXED with a one- / two-edge code (e[4]), its denominator is certified like UNIT's, the
// footprint is whole, and its region is STAGED | BORDER (below): every tap of every COVERED pixel lies in the window or on the one
// ring of pixels around the frame that the kernel paints into the window in the border colour -- so these footprints (the ring along
// the frame border of a stabilised clip, ~3 % of all) run the staged gather instead of clamped taps + selects.
#define MF_PLAN_BORDER 0x1000u          // (bit 12: clear in the raw row / column numbers of an overflow list, like bit 13)
// A list of exactly TWO cells leaves entries 2 and 3 unused; MF_PLAN_HOT in entry 2 then certifies the "pair" shape: the first
// (later, winning) cell is MIXED with ONE mask edge that can fail inside the footprint (its code in e[4]); whatever it does not take
// belongs to the second cell (which is IN, or single-edge with the pair covering every pixel); both denominators stay in
// (0.52, 1.9); the footprint is whole and its region STAGED and DEEP.  The kernel decides ownership with one float32 edge function
// per pixel and never looks at the second cell's edges.
// A short list (2-4 cells) whose MIXED entries all have a one- or two-edge code carries MF_PLAN_HOT in its first edge-code word e[4]
// (bits 6-7 of that word: entries - 1) when the footprint is whole, its region STAGED and interior (every tap two pixels inside the
// frame) and every denominator in (0.52, 1.9): the "multi" path of the kernel -- ownership from the coded edges only, coverage
// checked at run time (a pixel without owner sends the wavefront to the general code).
// ... and in the same entry 2: MF_PLAN_PAIR_FAST = BOTH cells satisfy the premises of the cheap coordinate chain (MF_PLAN_FAST64's),
// so the kernel may take its lane-uniform form of the pair
// path; MF_PLAN_PAIR_VERT = the deciding edge is closer to vertical than to horizontal: transposed lanes (a lane = 1 column x 4 rows).
#define MF_PLAN_PAIR_FAST 0x0001u
#define MF_PLAN_PAIR_VERT 0x0002u
// ... and bit 12 of the first edge-code word e[4] (next to its MF_PLAN_HOT = the multi path): every listed cell satisfies those premises.
#define MF_PLAN_MULTI_FAST 0x1000u
#define MF_PLAN_COUNT_SHIFT 6
struct alignas(16) FootPlan { uint16_t e[8]; };
// Source region of a footprint (FootRegion).  STAGED: every bilinear tap of every pixel of the footprint lies in columns sx0 ..
// sx0+MF_STAGE_COLS-1 and rows sy0 .. sy0+MF_STAGE_ROWS-1 of the source frame, and rows sy0 .. sy0+MF_STAGE_ROWS are inside the
// frame.  The warp kernel then copies MF_STAGE_CHUNKS 16-byte chunks (rows of MF_STAGE_PITCH bytes starting at the dword that
// holds column sx0, byte bs = 3 sx0 & ~3 of the row) into LDS with two global->LDS loads per lane and reads the taps from there.
// What the kernel needs is stored ready-made (the scalar unit is as busy as the vector unit there):
//   flags_origin: bit 31 STAGED, bit 30 DEEP, bits 0-22 origin = MF_STAGE_PITCH sy0 + bs  (tap (ix, iy) sits at LDS byte
//                 MF_STAGE_PITCH iy + 3 ix - origin of the window)
//   src_dwords:   (row_bytes sy0 + bs) / 4, the window's first dword in the frame (a frame is below 4 GB)
struct alignas(8) FootRegion { uint32_t flags_origin, src_dwords; };
#define MF_REGION_ORIGIN_MASK 0x007FFFFFu
#define MF_STAGE_PITCH 160
#define MF_STAGE_ROWS 12
#define MF_STAGE_COLS 52
#define MF_REGION_STAGED 0x80000000u
// bit 30 = DEEP (only with STAGED): the candidate list ends with an IN cell, or consists of two single-edge cells whose masks
// provably overlap across the footprint (either way every pixel has an owner), and the region
// stays two pixels inside the frame, so the warp kernel needs neither the interior check nor the crop flags.
#define MF_REGION_DEEP 0x40000000u
// bit 29 = COMPACT (only on the certified shapes -- HOT, PAIR, MULTI footprints: whole, interior, STAGED, NOFLAG; DEEP unless the multi path has
// to check coverage): every tap of every pixel, whichever listed cell owns it, lies in MF_COMPACT_ROWS rows of MF_COMPACT_PITCH bytes
// starting at the dword that holds column sx0 of row sy0 -- 63 chunks of 16 bytes, ONE global->LDS load per wavefront; origin and
// src_dwords then refer to that layout (origin = MF_COMPACT_PITCH sy0 + bs).  The warp kernel maps lanes to footprint rows differently
// on such a window (rows 0, 2, 4, 6 in lanes 0-31: no LDS bank conflicts at this pitch, warp.hip).
#define MF_REGION_COMPACT 0x20000000u
#define MF_COMPACT_PITCH 112
// bit 28 = NOFLAG (with or without STAGED): no pixel of the footprint can pass one of the four crop-boundary tests of mfs.py:1075-1098
// -- every source coordinate any listed cell can give it lies more than one pixel inside the frame's first / last column and row --
// so the scan-only pass (crop_scan_kernel, warp.hip) skips it.  DEEP implies it.
#define MF_REGION_NOFLAG 0x10000000u
// bit 27 = BORDER (only with STAGED, never with DEEP): the window of a border-path footprint (MF_PLAN_BORDER).  Rows sy0 .. sy0 + 11 lie
// inside the frame (sy0 <= H - 12: the kernel does not fetch a 13th row for it); bits 23-26 say which pixels just OUTSIDE the frame the
// taps can reach and the kernel therefore paints in the border colour after the copy: column -1 (the three bytes in front of each
// window row; the window then starts at byte 0 of the frame rows and needs no column beyond 51), column W (the three bytes behind
// column W-1; the window then ends with the frame rows), row -1 (the LDS row in front of the window, sy0 = 0), row H (the row behind
// it, sy0 = H - 12).
#define MF_REGION_BORDER 0x08000000u
#define MF_REGION_PAINT_LEFT 0x04000000u
#define MF_REGION_PAINT_RIGHT 0x02000000u
#define MF_REGION_PAINT_TOP 0x01000000u
#define MF_REGION_PAINT_BOTTOM 0x00800000u
#define MF_COMPACT_ROWS 9
#define MF_STAGE_CHUNKS 128            // two 16-byte chunks per lane: 12 rows x 10 chunks + 8 chunks of a 13th row (unused)
struct TableView {
    double* records; CellBox* boxes; float* edges; float* uedges; FootPlan* plan; FootRegion* regions; int32_t* reach; int32_t* grid; int32_t* bounds;
};
inline size_t plan_count(int n, int W, int H)
{
    return (size_t)n * ((H + MF_FOOT_H - 1) / MF_FOOT_H) * ((W + MF_FOOT_W - 1) / MF_FOOT_W);
}
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
inline size_t plan_offset(int n, int R, int C)
{
    return align16(table_records(n, R, C) * (MF_CELL_DOUBLES * sizeof(double) + sizeof(CellBox) + (MF_EDGE_FLOATS + MF_UEDGE_FLOATS) * sizeof(float)));
}
inline size_t table_bytes(int n, int W, int H, int R, int C)
{
    return plan_offset(n, R, C) + plan_count(n, W, H) * (sizeof(FootPlan) + sizeof(FootRegion)) +
           (size_t)n * reach_parts(R, C) * 4 * sizeof(int32_t) + (size_t)(R + C + 2) * sizeof(int32_t) + 4 * sizeof(int32_t);
}
inline TableView table_view(void* blob, int n, int W, int H, int R, int C);
// The part of a table that belongs to frames f0 ... (the per-frame sections advanced, the vertex grid shared): what launch_warp /
// launch_crop_scan need to work on a chunk of a clip whose table was built in one go.
inline TableView table_slice(const TableView& t, int f0, int W, int H, int R, int C)
{
    TableView v = t;
    const size_t rec = table_records(f0, R, C), fp = (size_t)f0 * ((H + MF_FOOT_H - 1) / MF_FOOT_H) * ((W + MF_FOOT_W - 1) / MF_FOOT_W);
    v.records += rec * MF_CELL_DOUBLES; v.boxes += rec; v.edges += rec * MF_EDGE_FLOATS; v.uedges += rec * MF_UEDGE_FLOATS;
    v.plan += fp; v.regions += fp; v.reach += (size_t)f0 * reach_parts(R, C) * 4;
    return v;
}
inline TableView table_view(void* blob, int n, int W, int H, int R, int C)
{
    const size_t nrec = table_records(n, R, C);
    TableView v;
    v.records = (double*)blob;
    v.boxes = (CellBox*)(v.records + nrec * MF_CELL_DOUBLES);
    v.edges = (float*)(v.boxes + nrec);
    v.uedges = v.edges + nrec * MF_EDGE_FLOATS;
    v.plan = (FootPlan*)((char*)blob + plan_offset(n, R, C));
    v.regions = (FootRegion*)(v.plan + plan_count(n, W, H));
    v.reach = (int32_t*)(v.regions + plan_count(n, W, H));
    v.grid = v.reach + (size_t)n * reach_parts(R, C) * 4;
    v.bounds = v.grid + (R + C + 2);
    return v;
}

// n / d for n < 2^31 as (n * m) >> (32 + s), m = ceil(2^(31 + ceil(log2 d)) / d): exact because the excess m d - 2^p is below
// d <= 2^(p - 31).  d == 1 is passed through.
struct FastDiv {
    uint32_t d, m, s;
    __device__ __forceinline__ uint32_t quotient(uint32_t n) const { return d == 1u ? n : (__umulhi(n, m) >> s); }
};
inline FastDiv make_fast_div(uint32_t d)
{
    FastDiv f;
    f.d = d; f.m = 0; f.s = 0;
    if (d > 1u) {
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;                       // ceil(log2 d) >= 1
        const unsigned p = 31u + l;
        f.m = (uint32_t)(((1ull << p) + d - 1) / d);
        f.s = l - 1u;
    }
    return f;
}
// XCD-aware tile order (workgroups go to the 8 XCDs round-robin by linear id, each XCD has its own L2): workgroup L of a
// 1-D grid of 8 * per_xcd takes tile (L % 8) * per_xcd + L / 8, so every XCD sweeps one contiguous eighth of the tiles
// in raster order and rows shared by neighbouring tiles are fetched once per L2.  tile -> (frame, tile row, tile col).
struct TileOrder {
    uint32_t tiles, per_xcd;
    FastDiv by_frame, by_row;
    __device__ __forceinline__ bool decode(uint32_t block, int& f, int& ty, int& tx) const
    {
        const uint32_t tile = (block & 7u) * per_xcd + (block >> 3);
        if (tile >= tiles) return false;
        f = (int)by_frame.quotient(tile);
        const uint32_t in_frame = tile - (uint32_t)f * by_frame.d;
        ty = (int)by_row.quotient(in_frame);
        tx = (int)(in_frame - (uint32_t)ty * by_row.d);
        return true;
    }
};
inline bool make_tile_order(int tiles_x, int tiles_y, int n, TileOrder& o)
{
    const long long tiles = (long long)tiles_x * tiles_y * n;
    if ((tiles + 7) / 8 * 8 > 0x7FFFFFFFll) return false;
    o.tiles = (uint32_t)tiles;
    o.per_xcd = (uint32_t)((tiles + 7) / 8);
    o.by_frame = make_fast_div((uint32_t)(tiles_x * tiles_y));
    o.by_row = make_fast_div((uint32_t)tiles_x);
    return true;
}



// Launch constants of the warp kernel (host-made: the kernel's scalar unit has none to spare).  Grid = (8 * per_xcd, frames): the
// workgroups of a frame go to the 8 XCDs round-robin by blockIdx.x, workgroup L takes footprint (L % 8) * per_xcd + L / 8, so
// every XCD sweeps one contiguous eighth of each frame in raster order.  All table offsets are 32-bit: launch_warp cuts a clip
// that would overflow them into several launches.
struct WarpGeom {
    uint32_t per_frame, per_xcd, nfx;            // footprints per frame, per XCD and frame, per row
    uint32_t div_m, div_s, div_pass;             // t / nfx = (mulhi(t, div_m) + (t & div_pass)) >> div_s
    uint32_t frame_bytes, row_bytes;             // 3 W H, 3 W
    uint32_t rec_frame_bytes, edge_frame_bytes;  // R C records / edge sets of one frame
    uint32_t cell_mul_x, cell_mul_y, mesh_cols, cell_last;   // the cell under a pixel of the UNWARPED grid: (mulhi(x, cell_mul_x), mulhi(y, cell_mul_y));
                                                 // row length C; R C - 1 (the speculative matrix load of the hot path, warp.hip)
};

// Launchers (defined next to their kernels).
int launch_jacobi(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                  int F, int S, int omega, int iters, hipStream_t st);
int launch_cell_table(const double* unstab, const double* stab, int n, int W, int H, int R, int C,
                      const TableView& tv, int32_t* crop, int32_t* status, hipStream_t st, bool first_of_table = true);
// (first_of_table: this launch also resets the table's clip-level rectangle -- false for the later frame ranges of a table that is
// built in several launches)
int launch_warp(const uint8_t* frames, uint8_t* out, const TableView& tv, int n, int W, int H, int R, int C,
                uint32_t border, int32_t* crop, hipStream_t st);
int launch_crop_scan(const TableView& tv, int n, int W, int H, int R, int C, int32_t* crop, hipStream_t st);
int check_d16_zero_fill(hipStream_t st);          // warp.hip: one-time device check the byte-tap kernels rely on
int launch_selftest_recip(unsigned long long n, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t st);
int launch_selftest_fast64(unsigned long long n, unsigned long long seed, unsigned long long* d_counters, hipStream_t st, unsigned long long* d_margin = nullptr);
size_t crop_resize_workspace_bytes(int W, int H);
int launch_crop_resize(const uint8_t* frames, uint8_t* out, int n, int W, int H, int left, int top, int right,
                       int bottom, void* work, hipStream_t st);
size_t vertex_motion_workspace_bytes(int total_features, int max_per_pair, int P, int R, int C);
int launch_vertex_motion(const double* early, const double* late, const int32_t* offsets, const double* hom, int P,
                         int total_features, int max_per_pair, int W, int H, int R, int C, int ell_rows, int ell_cols,
                         float* vel, double* disp, void* work, int32_t* status, hipStream_t st);
int launch_selftest_sqrt(unsigned long long n, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t st);
int launch_stability_score(const double* stab, int F, int S, double* ratio, double* score, hipStream_t st);
int launch_crop_reduce(const int32_t* crop, int n, int W, int H, int32_t* bounds, hipStream_t st);

}  // namespace mf

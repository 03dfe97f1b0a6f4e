// mf_warp_clip_u8c3: _get_stabilized_frames_and_crop_boundaries (mfs.py:909-1108) for a clip that is RESIDENT in HBM, as one call whose
// stages overlap INSIDE the clip.
//
// What depends on what: the cell table + footprint plan of a frame need its vertex displacements, nothing else; the warp of a frame
// needs that frame's table; the crop-boundary scan (mfs.py:1075-1098) and the clip-level rectangle (mfs.py:1103-1106) need the tables
// only (mf_crop_scan_f64) -- not the pixels.  So the clip is cut into `chunks` frame ranges and
//
//   prep stream:  table+plan(0) | table+plan(1) | ... | table+plan(chunks-1) | crop scan (all frames) | clip rectangle
//   stream:       ............... warp(0) ........... | warp(1) ............. | ... | warp(chunks-1)
//
// warp(k) waits for table+plan(k) only: after the first chunk's table the warp kernel never waits again, the remaining tables, the
// scan and the rectangle run beside it, and the rectangle (d_bounds) is final on the prep stream long before the last warp ends --
// a sharded run issues its 16-byte all-reduce there, beside the warp.  Five launches per chunk-step became two (the reach reset went
// into the cell-table kernel, the reduction off the critical path).
// The prep stream is the caller's (e.g. the one its Jacobi sweep runs on: the tables then simply follow the sweep) or, when NULL, an
// internal per-device stream forked from `stream` at the call.
#include <mutex>

#include "mf_common.h"

namespace mf {
namespace {

constexpr int CLIP_MAX_CHUNKS = 32;
struct ClipSide {
    std::mutex lock;
    int state = 0;                         // 0 unknown, 1 ready, -1 failed
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, done = nullptr, ready[CLIP_MAX_CHUNKS] = {};
};
ClipSide g_clip[64];

ClipSide* clip_side_for_current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    ClipSide& d = g_clip[dev];
    std::lock_guard<std::mutex> g(d.lock);
    if (d.state == 0) {
        d.state = -1;
        bool ok = hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&d.fork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&d.done, hipEventDisableTiming) == hipSuccess;
        for (auto& e : d.ready) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) d.state = 1;
    }
    return d.state == 1 ? &d : nullptr;
}

}  // namespace
}  // namespace mf

using namespace mf;

extern "C" int mf_warp_clip_u8c3(const uint8_t* d_frames, uint8_t* d_out, const double* d_unstab, const double* d_stab, int n, int W, int H,
                                 int R, int C, const uint8_t border_bgr[3], void* d_table, int32_t* d_crop, int32_t* d_bounds,
                                 int32_t* d_status, int chunks, void* prep_stream, void* stream)
{
    if (!d_frames || !d_out || !d_unstab || !d_stab || !border_bgr || !d_table || !d_crop || !d_bounds || !d_status) {
        set_error("mf_warp_clip_u8c3: null pointer");
        return MF_ERR_INVALID_ARG;
    }
    if (d_frames == d_out) { set_error("mf_warp_clip_u8c3: d_frames and d_out alias"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0 || W < 2 || H < 2) { set_error("mf_warp_clip_u8c3: bad sizes"); return MF_ERR_INVALID_ARG; }
    ClipSide* side = clip_side_for_current_device();
    if (!side) { set_error("mf_warp_clip_u8c3: no stream / events on this device"); return MF_ERR_HIP; }
    if (chunks <= 0) {
        // IN ORDER (the default of the product pipeline -- measured: kernels that run beside the warp kernel cost it more than they take
        // alone): the whole table on `stream`, then the warp by itself.  The rectangle comes from the warp's own fused
        // scan (folded into d_bounds by the kernel); or, when the caller gives a prep stream of its own, EARLY from the table (crop scan
        // there, beside the first microseconds of the warp) -- what a sharded run wants: its all-reduce then hides behind the warp as well.
        const hipStream_t st = (hipStream_t)stream;
        const hipStream_t prep = (hipStream_t)prep_stream;
        const uint32_t border = (uint32_t)border_bgr[0] | ((uint32_t)border_bgr[1] << 8) | ((uint32_t)border_bgr[2] << 16);
        TableView tv = table_view(d_table, n, W, H, R, C);
        tv.bounds = d_bounds;                  // the kernels fold the clip-level rectangle into the caller's 16 bytes themselves: no reduction launch
        if (const int rc = launch_cell_table(d_unstab, d_stab, n, W, H, R, C, tv, d_crop, d_status, st)) return rc;
        const bool early = prep != nullptr && prep != st;
        std::unique_lock<std::mutex> g(side->lock, std::defer_lock);
        if (early) {
            g.lock();
            MF_HIP_TRY(hipEventRecord(side->fork, st));
            MF_HIP_TRY(hipStreamWaitEvent(prep, side->fork, 0));
            if (const int rc = launch_crop_scan(tv, n, W, H, R, C, d_crop, prep)) return rc;
            MF_HIP_TRY(hipEventRecord(side->done, prep));
        }
        if (const int rc = launch_warp(d_frames, d_out, tv, n, W, H, R, C, border, d_crop, st)) return rc;
        if (early) MF_HIP_TRY(hipStreamWaitEvent(st, side->done, 0));
        return MF_OK;
    }
    if (chunks > CLIP_MAX_CHUNKS) chunks = CLIP_MAX_CHUNKS;
    if (chunks > n) chunks = n;
    const hipStream_t st = (hipStream_t)stream;
    hipStream_t prep = (hipStream_t)prep_stream;
    const uint32_t border = (uint32_t)border_bgr[0] | ((uint32_t)border_bgr[1] << 8) | ((uint32_t)border_bgr[2] << 16);
    TableView tv = table_view(d_table, n, W, H, R, C);
    tv.bounds = d_bounds;                      // (the first chunk's cell table sets the defaults there, the scan and the warps fold into it)
    const size_t vb1 = (size_t)(R + 1) * (C + 1) * 2, fb = (size_t)W * H * 3;
    const int per = (n + chunks - 1) / chunks;
    // the event set is per device: one call at a time records and waits on it (host side only -- the GPU work overlaps freely)
    std::lock_guard<std::mutex> g(side->lock);
    const bool one_stream = prep_stream != nullptr && prep == st;      // the caller wants everything in order on one stream
    if (!prep_stream) {
        prep = side->s;
        MF_HIP_TRY(hipEventRecord(side->fork, st));
        MF_HIP_TRY(hipStreamWaitEvent(prep, side->fork, 0));
    }
    int nk = 0;
    for (int i0 = 0; i0 < n; i0 += per, ++nk) {
        const int m = n - i0 < per ? n - i0 : per;
        if (const int rc = launch_cell_table(d_unstab + vb1 * i0, d_stab + vb1 * i0, m, W, H, R, C, table_slice(tv, i0, W, H, R, C),
                                             d_crop + 4 * (size_t)i0, d_status, prep, i0 == 0)) return rc;
        if (!one_stream) MF_HIP_TRY(hipEventRecord(side->ready[nk], prep));
    }
    if (const int rc = launch_crop_scan(tv, n, W, H, R, C, d_crop, prep)) return rc;
    if (!one_stream) MF_HIP_TRY(hipEventRecord(side->done, prep));
    nk = 0;
    for (int i0 = 0; i0 < n; i0 += per, ++nk) {
        const int m = n - i0 < per ? n - i0 : per;
        if (!one_stream) MF_HIP_TRY(hipStreamWaitEvent(st, side->ready[nk], 0));
        if (const int rc = launch_warp(d_frames + fb * i0, d_out + fb * i0, table_slice(tv, i0, W, H, R, C), m, W, H, R, C, border,
                                       d_crop + 4 * (size_t)i0, st)) return rc;
    }
    if (!one_stream) MF_HIP_TRY(hipStreamWaitEvent(st, side->done, 0));          // `stream` order now also implies: d_bounds is final
    return MF_OK;
}

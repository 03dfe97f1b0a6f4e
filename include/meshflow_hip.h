/* meshflow_hip.h -- C ABI of libmeshflow_hip.so (MI355X / gfx950).
 *
 * The reference (how4rd/meshflow, `mfs.py` = meshflowstabilizer.py) is pure Python and has no FFI layer;
 * its drop-in boundary is the pair of private methods called from stabilize() at mfs.py:150-158:
 *
 *   _get_stabilized_vertex_displacements        mfs.py:632-710   -> mf_jacobi_f64
 *   _get_stabilized_frames_and_crop_boundaries  mfs.py:909-1108  -> mf_cell_table_f64 + mf_warp_u8c3
 *
 * and the steps either side of them:
 *
 *   _crop_frames                                mfs.py:1111-1157 -> mf_crop_resize_u8c3
 *   vertex-motion accumulation after the tracker  mfs.py:268-282, 316-362, 365-452 -> mf_vertex_motion_f64
 *   _compute_stability_score                    mfs.py:1216-1259 -> mf_stability_score_f64
 *
 * Every entry point takes plain pointers and sizes.  Pointers named d_* are DEVICE pointers
 * (hipMalloc / torch tensors' data_ptr()); `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Kernel entry points are asynchronous on `stream`.  Return value: 0 on success,
 * a negative MF_ERR_* otherwise; mf_last_error() gives a thread-local message.  No CPU fallback
 * exists: without a GPU every compute entry point fails with MF_ERR_HIP.
 *
 * The Python binding is meshflow_amd/_lib.py (ctypes); INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 */
#ifndef MESHFLOW_HIP_H
#define MESHFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_ABI_VERSION 1

#define MF_OK 0
#define MF_ERR_INVALID_ARG (-1)   /* bad size / null pointer / unsupported shape */
#define MF_ERR_HIP (-2)           /* a HIP runtime call failed (see mf_last_error) */
#define MF_ERR_DEGENERATE (-3)    /* a mesh cell has no homography (cv2.findHomography would return None) */

/* Per-cell record written by mf_cell_table_f64 and read by mf_warp_u8c3: MF_CELL_DOUBLES float64.
 *   [0..8]   M    = inverse of the unstabilized->stabilized homography (what cv2.warpPerspective
 *                   evaluates, mfs.py:1041, 1052)
 *   [9..17]  Hi   = stabilized->unstabilized homography (cv2.perspectiveTransform, mfs.py:1042, 1054)
 *   [18..21] rect = L, T, Rt, B: inclusive pixel rect of the unstabilized cell (mfs.py:1045-1048)
 *   [22..25] bbox = x0, y0, x1, y1: inclusive, frame-clamped box outside which the cell's warped mask
 *                   is certainly zero (x0 > x1: empty)
 *   [26]     status: 0 ok, 1 degenerate
 *   [27..31] reserved
 * The table blob of n frames (mf_cell_table_bytes) holds n*R*C records followed by private acceleration
 * data of the warp kernel (compact boxes, edge functions, per-footprint candidate plan and source region, per-frame
 * reach, vertex grid). */
#define MF_CELL_DOUBLES 32
#define MF_CELL_OFF_M 0
#define MF_CELL_OFF_HI 9
#define MF_CELL_OFF_RECT 18
#define MF_CELL_OFF_BBOX 22
#define MF_CELL_OFF_STATUS 26

int mf_abi_version(void);
const char* mf_last_error(void);

/* ---- device plumbing (for hosts that do not bring their own allocator) ---- */
int mf_device_count(int* count);
int mf_set_device(int device);                            /* + the one-time device check of the byte-tap kernels */
int mf_malloc(void** d_ptr, size_t bytes);
int mf_free(void* d_ptr);
int mf_malloc_host(void** h_ptr, size_t bytes);          /* pinned host memory */
int mf_free_host(void* h_ptr);
int mf_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int mf_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int mf_stream_synchronize(void* stream);

/* ---- kernel 1: Jacobi temporal smoothing (mfs.py:844-878 for every vertex, mfs.py:695-704) ----
 * d_b, d_x: [F][S] float64, frame-major, S = (R+1)*(C+1)*2 independent series (the layout of the
 * reference's (F, R+1, C+1, 2) arrays).  x_start = b (mfs.py:699-703).  `iters` true Jacobi sweeps of
 *   x_new[t] = inv_on[t] * (b[t] + 2*lam[t] * sum_{d=-omega..omega, 0<=t+d<F} taps[d+omega]*x[t+d])
 * i.e. x <- diag(1/on) (b - off x) with off[t,t+d] = -2*lam[t]*taps[d+omega] (band includes d = 0,
 * mfs.py:767-781).  d_taps: [2*omega+1], d_lam, d_inv_on: [F].  d_b and d_x may not alias.
 * Any F, any omega (as the reference: mfs.py:193-213 reads every frame of the file): clips of up to 9,728 frames are swept with the
 * whole time axis of a series in LDS; longer ones in time tiles with a halo of (sweeps per launch) x omega frames -- bit-identical -- which
 * takes ceil(iters / k) launches and, from the second launch on, ONE scratch array of F*S doubles (hipMallocAsync / hipFreeAsync on
 * `stream`); radii beyond 246 on such clips go sweep by sweep through global memory.  Every output sums its taps in ascending order from
 * zero with one fma each, whatever the form. */
int mf_jacobi_f64(const double* d_b, double* d_x, const double* d_taps, const double* d_lam,
                  const double* d_inv_on, int F, int S, int omega, int iters, void* stream);

/* ---- kernel 2a: per-cell homography table (mfs.py:881-906, 964-967, 1025-1027, 1039-1048) ----
 * d_unstab, d_stab: [n][(R+1)*(C+1)][2] float64 vertex displacements of the n frames to warp.
 * d_table: mf_cell_table_bytes(n, W, H, R, C) bytes, 16-byte aligned.  d_crop: [n][4] int32, initialised here to the
 * per-frame defaults {0, 0, W-1, H-1} = {left, top, right, bottom} (mfs.py:992-995).
 * d_status: one int32, incremented once per degenerate cell (zero it before the call). */
size_t mf_cell_table_bytes(int n, int W, int H, int R, int C);
/* Byte offset, inside the table blob, of the CLIP-LEVEL rectangle of the table's n frames: 4 int32 {max left, max top, min right,
 * min bottom} (mfs.py:1103-1106).  mf_cell_table_f64 sets it to the defaults {0, 0, W-1, H-1}; every mf_warp_u8c3 / mf_crop_scan_f64 on
 * the table folds its frames' values into it next to the per-frame rows of d_crop -- after the warp (or the scan) of all n frames it
 * equals what mf_crop_reduce computes from d_crop, without that launch. */
size_t mf_cell_table_bounds_offset(int n, int W, int H, int R, int C);
int mf_cell_table_f64(const double* d_unstab, const double* d_stab, int n, int W, int H, int R, int C,
                      void* d_table, int32_t* d_crop, int32_t* d_status, void* stream);

/* ---- kernel 2b: mesh warp + crop-boundary scan (mfs.py:1017-1019, 1050-1098) ----
 * d_frames, d_out: [n][H][W][3] uint8 (BGR).  For every output pixel: owner = last cell in row-major
 * order whose warped mask is non-zero (mfs.py:1060-1061); source coordinates from that cell's Hi
 * (mfs.py:1054); cv2.remap bilinear, constant border colour (mfs.py:1063-1069); the four edge scans
 * of mfs.py:1075-1098 are folded into d_crop[f] = {left, top, right, bottom} with atomic max/min.
 * d_frames and d_out may not alias. */
int mf_warp_u8c3(const uint8_t* d_frames, uint8_t* d_out, const void* d_table, int n, int W, int H,
                 int R, int C, const uint8_t border_bgr[3], int32_t* d_crop, void* stream);

/* ---- the crop-boundary scan WITHOUT the pixels (mfs.py:1075-1098) ----
 * The four per-frame edge scans look at the coordinate maps only, i.e. at nothing but the cell table: this fills
 * d_crop[f] = {left, top, right, bottom} for the n frames of d_table exactly as mf_warp_u8c3 does (same ownership and coordinate
 * arithmetic, atomic max/min into the defaults mf_cell_table_f64 wrote), visiting only the footprints that can set a flag (a few
 * per cent: the ring along the frame border) and touching no frame.  With it the clip-level rectangle (mf_crop_reduce, and the
 * 16-byte all-reduce of a sharded clip) is known BEFORE the first pixel moves, so _crop_frames (mfs.py:159) can follow the warp
 * chunk by chunk.  Running mf_warp_u8c3 on the same d_crop afterwards changes nothing (max/min of equal values). */
int mf_crop_scan_f64(const void* d_table, int n, int W, int H, int R, int C, int32_t* d_crop, void* stream);

/* ---- the same three calls with the clip-level rectangle in the CALLER's memory ----
 * mf_cell_table_f64 / mf_warp_u8c3 / mf_crop_scan_f64 with d_bounds[4] int32 {max left, max top, min right, min bottom}
 * (mfs.py:1103-1106) in place of the four words inside the table blob (mf_cell_table_bounds_offset): the cell table sets the defaults
 * {0, 0, W-1, H-1} there, the warp / the scan fold their frames' values into it.  A pipeline that reuses one table for clip after clip
 * gives every clip its own 16 bytes: a rectangle handed to a consumer is never rewritten by a later clip. */
int mf_cell_table_bounds_f64(const double* d_unstab, const double* d_stab, int n, int W, int H, int R, int C,
                             void* d_table, int32_t* d_crop, int32_t* d_status, int32_t* d_bounds, void* stream);
int mf_warp_bounds_u8c3(const uint8_t* d_frames, uint8_t* d_out, const void* d_table, int n, int W, int H,
                        int R, int C, const uint8_t border_bgr[3], int32_t* d_crop, int32_t* d_bounds, void* stream);
int mf_crop_scan_bounds_f64(const void* d_table, int n, int W, int H, int R, int C, int32_t* d_crop, int32_t* d_bounds, void* stream);

/* ---- kernels 2a + 2b + the rectangle for a clip RESIDENT in HBM, overlapped inside the clip (csrc/clippipe.hip) ----
 * _get_stabilized_frames_and_crop_boundaries (mfs.py:909-1108) as ONE call: mf_cell_table_f64 + mf_crop_scan_f64 + mf_crop_reduce on a
 * PREP stream, mf_warp_u8c3 on `stream`, the clip cut into `chunks` frame ranges (1..32; 4 is a good value) so that warp(k) waits for
 * table(k) only and everything else runs beside the warp.  d_unstab / d_stab / d_frames / d_out / d_table / d_crop / d_status as for the
 * three calls it replaces (d_status is incremented, not reset); d_bounds: [4] int32 = the clip-level rectangle {max left, max top,
 * min right, min bottom} of these n frames, FINAL on the prep stream right after the tables -- long before the last warp ends (a
 * sharded run issues its 16-byte all-reduce there).  prep_stream: the caller's (work already queued on it -- e.g. the Jacobi sweep that
 * produces d_stab -- precedes the tables; `stream` is NOT waited for, so the caller orders reuse of d_table / d_crop itself), or NULL =
 * an internal per-device stream that starts after everything queued on `stream` so far; prep_stream == stream: everything in order on
 * one stream.  On return `stream` has been made to wait for the prep work: stream order implies d_out, d_crop and d_bounds are final.
 * chunks <= 0: IN ORDER -- the whole table on `stream`, then the warp alone, then (prep_stream NULL or == stream) the rectangle from the
 * warp's own scan by one reduction; with a prep stream of the caller's the rectangle is taken EARLY from the table there (crop scan +
 * reduction beside the start of the warp), so that a sharded run's all-reduce hides behind the warp.  This is the faster arrangement on
 * an MI355X (kernels running beside the warp kernel cost it more than they take alone); the chunked one gives the rectangle earliest. */
int mf_warp_clip_u8c3(const uint8_t* d_frames, uint8_t* d_out, const double* d_unstab, const double* d_stab, int n, int W, int H,
                      int R, int C, const uint8_t border_bgr[3], void* d_table, int32_t* d_crop, int32_t* d_bounds, int32_t* d_status,
                      int chunks, void* prep_stream, void* stream);

/* Clip-level crop bounds (mfs.py:1103-1106): {max left, max top, min right, min bottom} over n frames.
 * d_bounds: [4] int32. */
int mf_crop_reduce(const int32_t* d_crop, int n, int W, int H, int32_t* d_bounds, void* stream);

/* ---- next row on the path: crop + bilinear resize (mfs.py:1111-1157, called at mfs.py:159) ----
 * Crops each of the n frames to the inclusive rectangle {left, top, right, bottom} (the clip-level crop
 * bounds) and scales it back to W x H exactly like cv2.resize(crop, (W, H)) with the default INTER_LINEAR on
 * 8-bit data (two-pass 11-bit fixed point).  d_work: mf_crop_resize_workspace_bytes(W, H) bytes of scratch.
 * An empty or out-of-frame rectangle is MF_ERR_INVALID_ARG (cv2.resize fails on an empty source). */
size_t mf_crop_resize_workspace_bytes(int W, int H);
int mf_crop_resize_u8c3(const uint8_t* d_frames, uint8_t* d_out, int n, int W, int H, int left, int top, int right,
                        int bottom, void* d_work, void* stream);

/* ---- the row before the path: vertex-motion accumulation (mfs.py:236-452 from the matched features on) ----
 * Replaces the Python loops of _get_vertex_nearby_feature_residual_velocities (mfs.py:365-452), the medians, global
 * motion and median blur of _get_unstabilized_vertex_velocities (mfs.py:316-362, everything after the tracker call)
 * and the running sum of _get_unstabilized_vertex_displacements_and_homographies (mfs.py:268-282).  The tracker
 * itself (_get_matched_features_and_homography, mfs.py:455-629: FAST / LK / RANSAC) stays with the caller.
 * d_early, d_late: [total_features][2] float64 (x, y), the features of the P frame pairs back to back;
 * d_offsets: [P+1] int32, pair p owns features d_offsets[p] .. d_offsets[p+1]-1 (an empty range = no features);
 * max_per_pair >= every range length; d_hom: [P][9] float64 early-to-late homographies.
 * d_velocities: [P][(R+1)*(C+1)][2] float32 (what _get_unstabilized_vertex_velocities returns per pair);
 * d_displacements: [P+1][(R+1)*(C+1)][2] float64, [0] = 0 (mfs.py:271).
 * d_work: mf_vertex_motion_workspace_bytes(...) bytes, 16-byte aligned.  d_status: one int32 (zero it before the
 * call); non-zero afterwards = the reference's math.sqrt would have raised ValueError (mfs.py:444). */
size_t mf_vertex_motion_workspace_bytes(int total_features, int max_per_pair, int P, int R, int C);
int mf_vertex_motion_f64(const double* d_early, const double* d_late, const int32_t* d_offsets, const double* d_hom,
                         int P, int total_features, int max_per_pair, int W, int H, int R, int C,
                         int ellipse_rows, int ellipse_cols, float* d_velocities, double* d_displacements,
                         void* d_work, int32_t* d_status, void* stream);

/* ---- stability score (mfs.py:1216-1259) of device-resident vertex paths ----
 * d_stab: [F][S] float64 as for mf_jacobi_f64 (S = V*2: x and y of every vertex interleaved).  Per series the fraction of
 * the velocity profile's spectral energy that sits in DFT bins 1..5 (five direct sums + Parseval for the total);
 * d_series: [S] float64 receives those fractions, d_score: [1] float64 the clip-level score
 * (mean over x series + mean over y series) / 2.  F >= 2; with fewer than 7 frames the slice [1:6] of mfs.py:1250-1251 holds
 * fewer bins (bins 1..min(5, F-2)), exactly as in the reference.  Agrees with np.fft to float64 rounding. */
int mf_stability_score_f64(const double* d_stab, int F, int S, double* d_series, double* d_score, void* stream);

/* Device self-test: sqrt() on (0, 0.25] (the ellipse half-width, mfs.py:444) must be correctly rounded; *mismatches
 * receives the number of inputs where it is not (must be 0).  Synchronous. */
int mf_selftest_sqrt(uint64_t n, uint64_t seed, uint64_t* mismatches);

/* Device self-test: the warp kernel's trimmed reciprocal (exact for 0.5 <= |w| <= 2) against IEEE 1.0/w on
 * n hashed inputs; *mismatches receives the number of differing bit patterns (must be 0). Synchronous. */
int mf_selftest_recip(uint64_t n, uint64_t seed, uint64_t* mismatches);

/* Device self-test: the warp kernel's cheap float64 coordinate chain + float32-midpoint guard (hot footprints)
 * against cv2.perspectiveTransform's own arithmetic on n hashed (matrix, position) cases that satisfy the plan's premises.
 * counters[0] = float32 coordinates that differ from the exact chain's WITHOUT the guard raising its flag (must be 0),
 * counters[1] = values the guard flagged (each sends its wavefront to the exact chain), counters[2] = values tested.
 * Synchronous. */
int mf_selftest_fast64(uint64_t n, uint64_t seed, uint64_t counters[3]);

/* ... and how far the cheap chain's float64 values lie from the exact chain's on the same cases: *max_ulps receives the largest
 * distance in float64 ulps (the certified bound is 118, the guard's window 512).  Synchronous. */
int mf_selftest_fast64_margin(uint64_t n, uint64_t seed, double* max_ulps);

/* ---- host-buffer convenience wrappers (synchronous; H2D, kernels, D2H on an internal stream) ----
 * These are what a ctypes stub inside the reference's two methods would call.
 * kernel_ms (optional) receives the device time of the kernels alone, measured with HIP events.
 * The warp wrappers do NOT work in place: input and output frames may not overlap in memory (MF_ERR_INVALID_ARG), also for the
 * contiguous mf_warp_u8c3_host.  On any error return the contents of the output buffers are undefined (a failure that is known
 * before the first frame moves -- bad arguments, a degenerate mesh or an empty rectangle in the crop variant -- leaves them untouched). */
int mf_jacobi_f64_host(const double* b, double* x, const double* taps, const double* lam,
                       const double* inv_on, int F, int S, int omega, int iters, float* kernel_ms);
int mf_warp_u8c3_host(const uint8_t* frames, uint8_t* out, const double* unstab, const double* stab,
                      int n, int W, int H, int R, int C, const uint8_t border_bgr[3],
                      int32_t* crop /* [n][4] */, float* kernel_ms);
/* The same for frames that are separate allocations (the reference's Python lists of per-frame arrays, mfs.py:997, 1100):
 * frames[i] / out[i] point to frame i, H*W*3 bytes each.  mf_warp_u8c3_host is this with frames[i] = frames + i*H*W*3.
 * Both move the clip in chunks of ~16 MB (3 frames at 1080p, 1 at 4K) on four upload and four download threads with their own
 * HIP streams (each takes the next chunk when it is free; plus eight threads that fault the output pages in ahead of the downloads)
 * through a RING of ~720 MB of chunk buffers per direction (40 slots at 1080p, 30 at 4K) -- device memory is O(chunk) whatever the
 * length of the clip (1.5 GB of ring, ~1.75 GB in all at the peak, for any 1080p or 4K clip); a chunk is warped
 * as soon as it has landed (its cell table + plan are built right in front of its warp) and travels back while later chunks are still
 * going up (pageable memory is fine; memory from mf_malloc_host makes the copies truly asynchronous).  MF_PIPE_CHUNK (frames per
 * chunk) / MF_PIPE_SLOTS / MF_PIPE_UP / MF_PIPE_DOWN / MF_PIPE_POPULATE in the environment retune it
 * (read at every call; MF_PIPE_TRACE=1 prints the call's wall-clock milestones on stderr, 2 also every chunk's copy intervals).  Input and output frames must NOT overlap in memory (MF_ERR_INVALID_ARG): output pages are touched
 * while the input is still being read.  Device buffers and streams are kept between calls, grow-only, ONE CACHE PER DEVICE
 * (the calling thread's current device, mf_set_device): calls on one device are serialised, calls on different devices
 * from different host threads run concurrently; mf_host_cache_release() frees all of them. */
int mf_warp_u8c3_host_frames(const uint8_t* const* frames, uint8_t* const* out, const double* unstab, const double* stab,
                             int n, int W, int H, int R, int C, const uint8_t border_bgr[3],
                             int32_t* crop /* [n][4] */, float* kernel_ms);
/* ... followed by the next step of stabilize(), _crop_frames (mfs.py:159, 1111-1157), in the same pipeline: before any frame
 * moves, the cell tables of the whole clip are built (piece by piece, into one scratch table) and the clip-level rectangle
 * {max left, max top, min right, min bottom} (mfs.py:1103-1106) taken from them (mf_crop_scan_f64 + mf_crop_reduce: the edge scans look
 * at the coordinate maps only) and written to
 * bounds[4]; then every chunk goes up, is warped, cropped to the rectangle and resized back to W x H (mf_crop_resize_u8c3) and comes
 * down in ONE phase, both PCIe directions busy throughout: cropped[i] receives frame i of what stabilize() hands to the encoder.
 * `out` (the uncropped stabilized frames) may be NULL: they then never cross PCIe.  A degenerate mesh (MF_ERR_DEGENERATE) or an empty
 * rectangle (MF_ERR_INVALID_ARG: cv2.resize fails on an empty source) ends the call before any frame is uploaded or any output byte
 * written; bounds[] and crop[] are written in the second case. */
int mf_warp_crop_u8c3_host_frames(const uint8_t* const* frames, uint8_t* const* out /* may be NULL */, uint8_t* const* cropped,
                                  const double* unstab, const double* stab, int n, int W, int H, int R, int C,
                                  const uint8_t border_bgr[3], int32_t* crop /* [n][4] */, int32_t bounds[4], float* kernel_ms);
/* _crop_frames (mfs.py:1111-1157, called at mfs.py:159) BY ITSELF on host frames: frames[i] (H*W*3 bytes each) are cropped to the
 * inclusive rectangle {left, top, right, bottom} and scaled back to W x H (mf_crop_resize_u8c3) into cropped[i], through the same
 * ring of chunk buffers and copy threads as the warp wrappers (upload, resize, download of the chunks all overlap).  An empty or
 * out-of-frame rectangle is MF_ERR_INVALID_ARG before any output byte is written. */
int mf_crop_resize_u8c3_host_frames(const uint8_t* const* frames, uint8_t* const* cropped, int n, int W, int H, int left, int top,
                                    int right, int bottom, float* kernel_ms);
int mf_host_cache_release(void);

/* ---- multi-GPU exchange steps, on RCCL directly: ONE process drives the GPUs 0..ndev-1 of a node ----
 * The path shards by contiguous frame range: every GPU warps its own frames (mf_warp_u8c3 on that device) after a replicated
 * mf_jacobi_f64; these are the only two exchanges.  librccl is opened with dlopen at mf_comm_init_all (error -2 when the
 * box has none).  RCCL errors come back as -1000 - ncclResult_t.  All three calls are synchronous; the device buffers
 * handed in must be complete (mf_stream_synchronize the streams that wrote them first).
 *   mf_comm_init_all(ndev)   ncclCommInitAll over devices 0..ndev-1 (rccl.h:236) + one stream per device
 *   mf_allreduce_crop        d_bounds[g]: {left, top, right, bottom} int32 on device g (that shard's mf_crop_reduce result);
 *                            afterwards every device holds the clip-level rectangle {max, max, min, min} (mfs.py:1103-1106):
 *                            16 bytes, one grouped call (ncclAllReduce max on the first pair, min on the second)
 *   mf_gather_frames         d_shards[g] (shard_bytes[g] bytes on device g, the stabilized frames of rank g's frame range) ->
 *                            d_dst on device `root`, back to back in rank order: one group of ncclSend / ncclRecv with
 *                            per-rank byte counts (shards are ragged when ndev does not divide the frame count)
 *   mf_comm_destroy          frees the communicators and streams
 * STATUS: exercised with ONE rank only (tests/test_gpu_nccl_one_rank.py; no multi-GPU node was ever available to the build).  It is
 * the exchange for a single-process C/C++ host that binds this library directly; the Python product and
 * `bench.py --gpus N` do NOT use it -- their one exchange implementation is meshflow_amd/dist.py (one process per GPU,
 * torch.distributed: "nccl" = RCCL), covered at world_size 2 and 8 under gloo (tests/test_dist_gloo.py, tests/test_dist_gloo8.py). */
int mf_comm_init_all(int ndev);
int mf_comm_size(int* ndev);
int mf_allreduce_crop(int32_t* const* d_bounds);
int mf_gather_frames(const uint8_t* const* d_shards, const size_t* shard_bytes, uint8_t* d_dst, int root);
int mf_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* MESHFLOW_HIP_H */

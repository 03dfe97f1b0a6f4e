// Host-buffer entry points of the warp (what a ctypes stub inside the reference's
// _get_stabilized_frames_and_crop_boundaries, mfs.py:909-1108, calls): frames in host memory in, stabilized frames in host
// memory out, with the PCIe transfers chunked and overlapped with each other and with the kernels.
//
// The clip moves in chunks of ~16 MB (3 frames at 1080p, 1 at 4K; MF_PIPE_CHUNK = frames per chunk overrides) through a RING of
// ~720 MB of chunk-sized device buffers per direction (MF_PIPE_SLOTS overrides the slot count): device memory is O(chunk), whatever the
// length of the clip (2,000 frames of 1080p and all 1,200 frames of config 4's 4K clip go through 1.5 GB of ring).  UP host threads each
// own a HIP stream and copy the next chunk that has not gone up yet into the ring slot chunk k % slots -- once the download of the chunk that used the slot before has ended (an event
// wait on the upload stream, no host synchronisation) -- (pageable memory is fine: the runtime stages it; pinned memory --
// mf_malloc_host -- makes the copies truly asynchronous); the calling thread waits for chunk k's upload event, launches the cell table
// + plan + warp of that chunk on the compute stream and records an event; DOWN host threads wait for it on their own streams and copy
// the stabilized chunk back, while later chunks are still going up.  PCIe carries both directions at once and the kernels disappear
// behind the copies.  The ring, the per-chunk cell table, streams and small buffers are kept between calls (grow-only cache, ONE PER
// DEVICE, each serialised by its own mutex: a process that drives several GPUs from several host threads -- mf_set_device(g) per
// thread -- runs their clips concurrently; mf_host_cache_release frees them).
//
// With `cropped` the call also runs the next step of stabilize(), _crop_frames (mfs.py:159, 1111-1157), in the SAME single phase: the
// clip-level crop rectangle (mfs.py:1103-1106) is taken from the cell tables alone before the first frame moves (the tables of the
// whole clip are built piece by piece into one scratch table and scanned, mf_crop_scan_f64: ~1 ms per 1,000 frames), so every chunk is
// uploaded, warped, cropped + resized (into the ring slot its input came in: the warp was that slot's only reader) and downloaded
// while later chunks are still going up; `out` may then be NULL -- the reference only uses the cropped frames afterwards -- and the
// uncropped frames never cross PCIe.  mf_crop_resize_u8c3_host_frames is _crop_frames by itself through the same ring.
#include <stdlib.h>
#include <sys/mman.h>
#include <unistd.h>

#include <stdio.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "mf_common.h"

namespace mf {
namespace {

// Chunk size and ring depth (measured on MI355X hosts, profiles/r05_e2e_ring.txt and r05_pcie_duplex.txt).  A chunk is usable only when
// all of it has landed, so the downloads trail the uploads by (upload threads x chunk) at both ends of a clip -- with 50 MB chunks 4.5 ms
// of a 48 ms cfg2 clip at each end -- and small chunks win although their kernels are less efficient (they hide behind the copies):
// 1080p, interleaved repetitions, frames/s without | with the crop: 8 frames x 18 slots 5,330 | 5,510, 4 x 36 5,480 | 6,040,
// 3 x 48 5,880 | 6,030, 2 x 64 5,660 | 5,910;  4K 2 x 18 1,570 | 1,510, 1 x 36 1,655 | 1,545.
constexpr size_t PIPE_CHUNK_BYTES = (size_t)16 << 20;    // bytes per chunk: 3 frames at 1080p, 1 at 4K (MF_PIPE_CHUNK = frames overrides)
// (720 MiB: 40 slots at 1080p, 30 at 4K.  900 MiB were no faster and left the whole pipeline -- rings + a 92 MiB scan table + vertex paths,
// with a transient of up to ~190 MiB on top -- at 2.04-2.09 GiB of device memory; now 1.55 GiB steady, 1.73 at the peak.)
constexpr size_t PIPE_RING_BYTES = (size_t)720 << 20;    // ring per direction (MF_PIPE_SLOTS overrides the slot count), for a clip of any length
constexpr int PIPE_SLOTS_MAX = 64;
constexpr size_t PIPE_SCAN_TABLE_BYTES = (size_t)96 << 20;   // scratch table of the rectangle pre-pass: ~330 frames of 1080p at a 16 x 16 mesh per piece
constexpr int PIPE_UP = 4;         // upload threads / streams   (round 5, 16 MB chunks, frames/s at config 2: 3+3 5,730-6,080, 4+4 5,880-6,030, 6+6 5,070-5,100)
constexpr int PIPE_DOWN = 4;       // download threads / streams
constexpr int PIPE_POPULATE = 8;   // threads that fault the output pages in ahead of the downloads (MF_PIPE_POPULATE, 0 = none)
constexpr int PIPE_MAX = 8;        // upper bound on either thread count (MF_PIPE_UP / MF_PIPE_DOWN / MF_PIPE_CHUNK tune them)

int env_int(const char* name, int fallback, int lo, int hi)
{
    const char* v = getenv(name);
    if (!v || !*v) return fallback;
    const int x = atoi(v);
    return x < lo ? lo : (x > hi ? hi : x);
}

struct Grow {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t need(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr int PIPE_MAX_DEVICES = 64;
struct PipeCache {
    std::mutex lock;
    int device = -1;
    Grow frames, out, table, scan, unstab, stab, crop, status, work, bounds;       // frames / out: the two rings
    hipStream_t compute = nullptr, up[PIPE_MAX] = {}, down[PIPE_MAX] = {};
    void release()
    {
        frames.release(); out.release(); table.release(); scan.release(); unstab.release(); stab.release(); crop.release(); status.release();
        work.release(); bounds.release();
        if (compute) (void)hipStreamDestroy(compute);
        for (auto& s : up) { if (s) (void)hipStreamDestroy(s); s = nullptr; }
        for (auto& s : down) { if (s) (void)hipStreamDestroy(s); s = nullptr; }
        compute = nullptr;
        device = -1;
    }
};
PipeCache g_pipe[PIPE_MAX_DEVICES];          // one cache per device: nothing is freed or re-allocated when the current device changes

struct Shared {
    std::mutex m;
    std::condition_variable cv;
    std::vector<char> up_ready, warp_ready, populated, resize_ready, populated2, down_issued;
    hipError_t err = hipSuccess;
    const char* what = "";
    bool abort = false;
    void fail(hipError_t e, const char* w)
    {
        std::lock_guard<std::mutex> g(m);
        if (err == hipSuccess) { err = e; what = w; }
        abort = true;
        cv.notify_all();
    }
    void mark(std::vector<char>& v, int k)
    {
        { std::lock_guard<std::mutex> g(m); v[k] = 1; }
        cv.notify_all();
    }
    bool wait(std::vector<char>& v, int k)               // false: another thread failed
    {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return v[k] || abort; });
        return v[k] && !abort;
    }
};

// frames i0..i1-1 between host pointers and the device stack, as few copies as the host layout allows
// (`dev` = where frame i0 lives on the device: the chunk's ring slot)
hipError_t copy_frames(uint8_t* dev, const uint8_t* const* host, int i0, int i1, size_t fb, bool to_device, hipStream_t st)
{
    int i = i0;
    while (i < i1) {
        int j = i + 1;
        while (j < i1 && host[j] == host[j - 1] + fb) ++j;            // run of frames contiguous in host memory
        const size_t bytes = (size_t)(j - i) * fb;
        hipError_t e = to_device ? hipMemcpyAsync(dev + (size_t)(i - i0) * fb, host[i], bytes, hipMemcpyHostToDevice, st)
                                 : hipMemcpyAsync(const_cast<uint8_t*>(host[i]), dev + (size_t)(i - i0) * fb, bytes, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        i = j;
    }
    return hipSuccess;
}

// A freshly allocated output buffer (np.empty: untouched anonymous memory) costs one page fault + page clearing per 4 KB
// when the download threads first write it; taken inside the three or four download threads that roughly doubles the clip
// time (114 ms against 55 ms into a buffer that has been used before).  These threads fault the pages of chunk k in BEFORE
// chunk k's download may start (the download waits for a flag), several chunks in parallel and concurrently with the
// uploads: one byte written per page -- harmless, because the download overwrites every byte of the chunk afterwards.
// (Eight threads populate 1.87 GB in 12 ms on the MI355X hosts; MADV_POPULATE_WRITE was measured at half that rate.)
void populate_frames(uint8_t* const* host, int i0, int i1, size_t fb)
{
    static const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    int i = i0;
    while (i < i1) {
        int j = i + 1;
        while (j < i1 && host[j] == host[j - 1] + fb) ++j;
        const uintptr_t begin = (uintptr_t)host[i], end = begin + (size_t)(j - i) * fb;
        const uintptr_t lo = (begin + page - 1) & ~(page - 1), hi = end & ~(page - 1);
        for (uintptr_t a = lo; a < hi; a += page) *(volatile uint8_t*)a = 0;
        *(volatile uint8_t*)begin = 0;                                                  // the partial pages at both ends
        *(volatile uint8_t*)(end - 1) = 0;
        i = j;
    }
}

// Every host frame as an address interval; true when an interval of `a` overlaps one of `b` (the populate threads write into
// the output pages while the uploads are still reading the input: aliasing would corrupt the input silently).
bool ranges_overlap(const uint8_t* const* a, const uint8_t* const* b, int n, size_t fb)
{
    std::vector<std::pair<uintptr_t, int>> v;
    v.reserve(2 * (size_t)n);
    for (int i = 0; i < n; ++i) { v.emplace_back((uintptr_t)a[i], 0); v.emplace_back((uintptr_t)b[i], 1); }
    std::sort(v.begin(), v.end());
    uintptr_t end[2] = { 0, 0 };                       // furthest end seen so far per side
    for (const auto& it : v) {
        if (it.first < end[1 - it.second]) return true;
        const uintptr_t e = it.first + fb;
        if (e > end[it.second]) end[it.second] = e;
    }
    return false;
}

}  // namespace

// One job of the pipeline.  warp: frames -> (out: stabilized frames) (cropped: + _crop_frames of them); !warp: _crop_frames of `frames`
// by itself with the rectangle given (-> cropped).
struct PipeJob {
    const uint8_t* const* frames; uint8_t* const* out; uint8_t* const* cropped;
    const double* unstab; const double* stab;
    int n, W, H, R, C;
    uint32_t border;
    int32_t* crop; int32_t* bounds; float* kernel_ms;
    bool warp;
    int rect[4];
    const char* name;
};

int run_host_pipeline(PipeJob job)
{
    const uint8_t* const* frames = job.frames;
    uint8_t* const* out = job.out;
    uint8_t* const* cropped = job.cropped;
    const int n = job.n, W = job.W, H = job.H, R = job.R, C = job.C;
    float* kernel_ms = job.kernel_ms;
    int dev = 0;
    MF_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= PIPE_MAX_DEVICES) { set_error("%s: device %d out of range", job.name, dev); return MF_ERR_INVALID_ARG; }
    PipeCache& pc = g_pipe[dev];
    std::lock_guard<std::mutex> cache_guard(pc.lock);
    const size_t fb = (size_t)W * H * 3;
    if ((out && ranges_overlap(frames, out, n, fb)) || (cropped && ranges_overlap(frames, cropped, n, fb)) ||
        (out && cropped && ranges_overlap(out, cropped, n, fb))) {
        set_error("%s: input and output frames overlap in memory (in-place operation is not supported)", job.name);
        return MF_ERR_INVALID_ARG;
    }
    if (pc.device != dev) {                              // first use of this device: its streams
        hipError_t se = hipStreamCreateWithFlags(&pc.compute, hipStreamNonBlocking);
        for (auto& s : pc.up) if (se == hipSuccess) se = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : pc.down) if (se == hipSuccess) se = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (se != hipSuccess) { pc.release(); return hip_fail(se, "hipStreamCreateWithFlags (host pipeline)"); }
        pc.device = dev;
    }
    const size_t vb1 = (size_t)(R + 1) * (C + 1) * 2 * sizeof(double);        // vertex displacements of one frame
    // (read at every call: a host may retune between clips)
    const int by_bytes = (int)std::min<size_t>(4096, std::max<size_t>(1, (PIPE_CHUNK_BYTES + fb / 2) / fb));
    const int cfg_chunk = env_int("MF_PIPE_CHUNK", by_bytes, 1, 4096), cfg_up = env_int("MF_PIPE_UP", PIPE_UP, 1, PIPE_MAX),
              cfg_down = env_int("MF_PIPE_DOWN", PIPE_DOWN, 1, PIPE_MAX), cfg_pop = env_int("MF_PIPE_POPULATE", PIPE_POPULATE, 0, PIPE_MAX),
              cfg_slots = env_int("MF_PIPE_SLOTS", (int)std::min<size_t>(PIPE_SLOTS_MAX, std::max<size_t>(4, PIPE_RING_BYTES / (fb * cfg_chunk))), 2, PIPE_SLOTS_MAX);
    const int chunk = n < cfg_chunk ? n : cfg_chunk;
    const int nchunks = (n + chunk - 1) / chunk;
    const int slots = nchunks < cfg_slots ? nchunks : cfg_slots;
    const size_t slot_bytes = fb * chunk;
    MF_HIP_TRY(pc.frames.need(slot_bytes * slots)); MF_HIP_TRY(pc.out.need(slot_bytes * slots));
    if (job.warp) {
        MF_HIP_TRY(pc.unstab.need(vb1 * n)); MF_HIP_TRY(pc.stab.need(vb1 * n));
        MF_HIP_TRY(pc.table.need(table_bytes(chunk, W, H, R, C)));       // of ONE chunk: rebuilt in front of every chunk's warp (~20 us)
        MF_HIP_TRY(pc.crop.need((size_t)n * 4 * sizeof(int32_t)));
        MF_HIP_TRY(pc.status.need(sizeof(int32_t)));
        MF_HIP_TRY(pc.bounds.need(4 * sizeof(int32_t)));
    }
    if (cropped) MF_HIP_TRY(pc.work.need(crop_resize_workspace_bytes(W, H)));
    uint8_t* const ring_in = (uint8_t*)pc.frames.p;
    uint8_t* const ring_out = (uint8_t*)pc.out.p;
    int32_t* d_crop = (int32_t*)pc.crop.p;
    const double* d_unstab = (const double*)pc.unstab.p;
    const double* d_stab = (const double*)pc.stab.p;
    const size_t v2 = vb1 / sizeof(double);

    std::vector<hipEvent_t> up_done(nchunks, nullptr), warp_done(nchunks, nullptr), resize_done(nchunks, nullptr), down_done(nchunks, nullptr),
        t0(nchunks, nullptr), t1(nchunks, nullptr);
    hipEvent_t r0 = nullptr, r1 = nullptr;
    struct EventGuard {
        std::vector<hipEvent_t>* v[6];
        hipEvent_t *a, *b;
        ~EventGuard()
        {
            for (auto* vec : v) for (hipEvent_t e : *vec) if (e) (void)hipEventDestroy(e);
            if (*a) (void)hipEventDestroy(*a);
            if (*b) (void)hipEventDestroy(*b);
        }
    } guard{{&up_done, &warp_done, &resize_done, &down_done, &t0, &t1}, &r0, &r1};
    for (int k = 0; k < nchunks; ++k) {
        MF_HIP_TRY(hipEventCreateWithFlags(&up_done[k], hipEventDisableTiming));
        if (out) MF_HIP_TRY(hipEventCreateWithFlags(&warp_done[k], hipEventDisableTiming));
        if (cropped) MF_HIP_TRY(hipEventCreateWithFlags(&resize_done[k], hipEventDisableTiming));
        if (k + slots < nchunks) MF_HIP_TRY(hipEventCreateWithFlags(&down_done[k], hipEventDisableTiming));       // (only where a later chunk takes the slot over)
        if (kernel_ms) { MF_HIP_TRY(hipEventCreate(&t0[k])); MF_HIP_TRY(hipEventCreate(&t1[k])); }
    }
    if (kernel_ms) { MF_HIP_TRY(hipEventCreate(&r0)); MF_HIP_TRY(hipEventCreate(&r1)); }

    // Before any frame moves: vertex paths up and -- with `cropped` -- the clip-level crop rectangle from the cell tables alone
    // (mf_crop_scan_f64: the edge scans of mfs.py:1075-1098 look at the coordinate maps, not at pixels), so that _crop_frames can follow
    // every chunk's warp directly: the tables of the whole clip are built piece by piece into one scratch table (O(piece) memory) and
    // scanned.  A degenerate mesh or an empty rectangle (cv2.resize would fail on an empty source) ends the call here: no frame has
    // been uploaded, no output page touched.
    const uint32_t border = job.border;
    int32_t status = 0;
    int32_t rect[4] = { job.rect[0], job.rect[1], job.rect[2], job.rect[3] };
    if (job.warp) {
        MF_HIP_TRY(hipMemsetAsync(pc.status.p, 0, sizeof(int32_t), pc.compute));
        MF_HIP_TRY(hipMemcpyAsync(pc.unstab.p, job.unstab, vb1 * n, hipMemcpyHostToDevice, pc.compute));
        MF_HIP_TRY(hipMemcpyAsync(pc.stab.p, job.stab, vb1 * n, hipMemcpyHostToDevice, pc.compute));
    }
    if (kernel_ms) MF_HIP_TRY(hipEventRecord(r0, pc.compute));
    if (job.warp && cropped) {
        const size_t per_frame = table_bytes(1, W, H, R, C);
        int piece = (int)std::max<size_t>(1, PIPE_SCAN_TABLE_BYTES / per_frame);
        if (piece > n) piece = n;
        MF_HIP_TRY(pc.scan.need(table_bytes(piece, W, H, R, C)));
        for (int i0 = 0; i0 < n; i0 += piece) {
            const int m = n - i0 < piece ? n - i0 : piece;
            const TableView tv = table_view(pc.scan.p, m, W, H, R, C);
            if (const int rc0 = launch_cell_table(d_unstab + v2 * i0, d_stab + v2 * i0, m, W, H, R, C, tv, d_crop + 4 * (size_t)i0, (int32_t*)pc.status.p, pc.compute)) return rc0;
            if (const int rc0 = launch_crop_scan(tv, m, W, H, R, C, d_crop + 4 * (size_t)i0, pc.compute)) return rc0;
        }
        if (const int rc0 = launch_crop_reduce(d_crop, n, W, H, (int32_t*)pc.bounds.p, pc.compute)) return rc0;
    }
    if (kernel_ms) MF_HIP_TRY(hipEventRecord(r1, pc.compute));
    if (job.warp && cropped) {
        MF_HIP_TRY(hipMemcpyAsync(rect, pc.bounds.p, sizeof rect, hipMemcpyDeviceToHost, pc.compute));
        MF_HIP_TRY(hipMemcpyAsync(&status, pc.status.p, sizeof(int32_t), hipMemcpyDeviceToHost, pc.compute));
        MF_HIP_TRY(hipStreamSynchronize(pc.compute));
        if (job.bounds) for (int i = 0; i < 4; ++i) job.bounds[i] = rect[i];
        if (status != 0) {
            set_error("%s: %d degenerate mesh cell(s): no homography exists (cv2.findHomography would return None)", job.name, status);
            return MF_ERR_DEGENERATE;
        }
        if (rect[2] < rect[0] || rect[3] < rect[1]) {
            MF_HIP_TRY(hipMemcpy(job.crop, d_crop, (size_t)n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost));       // (the per-frame values say which frame emptied it)
            set_error("%s: empty crop rectangle (%d, %d, %d, %d): cv2.resize would fail on an empty source",
                      job.name, rect[0], rect[1], rect[2], rect[3]);
            return MF_ERR_INVALID_ARG;
        }
    }

    // MF_PIPE_TRACE=1: wall-clock milestones of the call on stderr (where a slow host loses its time: page population, uploads, downloads)
    const int trace_level = env_int("MF_PIPE_TRACE", 0, 0, 2);         // 2: + begin / end of every chunk's copy calls, printed after the call
    const bool trace = trace_level != 0;
    std::vector<double> tr_up0(trace_level == 2 ? nchunks : 0), tr_up1(tr_up0.size()), tr_dn0(tr_up0.size()), tr_dn1(tr_up0.size()), tr_k(tr_up0.size());
    const auto t_begin = std::chrono::steady_clock::now();
    const auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    Shared sh;
    std::atomic<int> next_up{0}, next_down{0};
    sh.up_ready.assign(nchunks, 0);
    sh.warp_ready.assign(nchunks, 0);
    sh.resize_ready.assign(nchunks, 0);
    sh.down_issued.assign(nchunks, 0);
    sh.populated.assign(nchunks, cfg_pop > 0 && out ? 0 : 1);
    sh.populated2.assign(nchunks, cfg_pop > 0 && cropped ? 0 : 1);
    std::vector<std::thread> workers;
    const int n_up = nchunks < cfg_up ? nchunks : cfg_up, n_down = nchunks < cfg_down ? nchunks : cfg_down;
    const int n_pop = nchunks < cfg_pop ? nchunks : cfg_pop;
    auto chunk_end = [&](int k) { return (k * chunk + chunk < n) ? k * chunk + chunk : n; };
    auto slot_in = [&](int k) { return ring_in + slot_bytes * (size_t)(k % slots); };
    auto slot_out = [&](int k) { return ring_out + slot_bytes * (size_t)(k % slots); };
    // where chunk k's cropped + resized frames are: with the warp in front, in the slot its input came in (the warp was that slot's only
    // reader); _crop_frames by itself writes the out ring
    auto slot_cropped = [&](int k) { return job.warp ? slot_in(k) : slot_out(k); };
    for (int t = 0; t < n_pop; ++t)
        workers.emplace_back([&, t] {
            for (int k = t; k < nchunks; k += n_pop) {       // chunk by chunk, in the order the downloads will need the pages
                if (out) { populate_frames(out, k * chunk, chunk_end(k), fb); sh.mark(sh.populated, k); }
                if (cropped) { populate_frames(cropped, k * chunk, chunk_end(k), fb); sh.mark(sh.populated2, k); }
                if (trace && (k == nchunks - 1 || k == 0)) fprintf(stderr, "[mf pipe] %8.2f ms  pages of chunk %d populated\n", since(), k);
            }
        });
    for (int t = 0; t < n_up; ++t)
        workers.emplace_back([&, t] {
            if (hipSetDevice(dev) != hipSuccess) { sh.fail(hipErrorInvalidDevice, "hipSetDevice (upload thread)"); return; }
            // the next chunk to whichever thread is free: the streams do not move data at the same rate (one of them typically runs a copy
            // in less than half the time the others take beside it, profiles/r05_pcie_duplex.txt) and the kernels take the chunks in order
            for (int k = next_up.fetch_add(1); k < nchunks; k = next_up.fetch_add(1)) {
                const int i0 = k * chunk, i1 = chunk_end(k);
                hipError_t e = hipSuccess;
                if (k >= slots) {                            // the ring slot is free once the chunk that used it before has gone down
                    if (!sh.wait(sh.down_issued, k - slots)) return;
                    e = hipStreamWaitEvent(pc.up[t], down_done[k - slots], 0);
                }
                if (trace_level == 2) tr_up0[k] = since();
                if (e == hipSuccess) e = copy_frames(slot_in(k), frames, i0, i1, fb, true, pc.up[t]);
                if (trace_level == 2) tr_up1[k] = since();
                if (e == hipSuccess) e = hipEventRecord(up_done[k], pc.up[t]);
                if (e != hipSuccess) { sh.fail(e, "upload of a frame chunk"); return; }
                sh.mark(sh.up_ready, k);
                if (trace && (k == nchunks - 1 || k < 2)) fprintf(stderr, "[mf pipe] %8.2f ms  upload of chunk %d issued\n", since(), k);
                { std::lock_guard<std::mutex> g(sh.m); if (sh.abort) return; }
            }
        });
    // Finished chunks travel back while later ones are still going up and being warped: ONE phase, both PCIe directions busy throughout.
    for (int t = 0; t < n_down; ++t)
        workers.emplace_back([&, t] {
            if (hipSetDevice(dev) != hipSuccess) { sh.fail(hipErrorInvalidDevice, "hipSetDevice (download thread)"); return; }
            for (int k = next_down.fetch_add(1); k < nchunks; k = next_down.fetch_add(1)) {
                if (out) {
                    if (!sh.wait(sh.warp_ready, k) || !sh.wait(sh.populated, k)) return;
                    hipError_t e = hipStreamWaitEvent(pc.down[t], warp_done[k], 0);
                    if (trace_level == 2) tr_dn0[k] = since();
                    if (e == hipSuccess) e = copy_frames(slot_out(k), out, k * chunk, chunk_end(k), fb, false, pc.down[t]);
                    if (trace_level == 2) tr_dn1[k] = since();
                    if (e != hipSuccess) { sh.fail(e, "download of a frame chunk"); return; }
                }
                if (cropped) {
                    if (!sh.wait(sh.resize_ready, k) || !sh.wait(sh.populated2, k)) return;
                    hipError_t e = hipStreamWaitEvent(pc.down[t], resize_done[k], 0);
                    if (trace_level == 2 && !out) tr_dn0[k] = since();
                    if (e == hipSuccess) e = copy_frames(slot_cropped(k), cropped, k * chunk, chunk_end(k), fb, false, pc.down[t]);
                    if (trace_level == 2) tr_dn1[k] = since();
                    if (e != hipSuccess) { sh.fail(e, "download of a cropped frame chunk"); return; }
                }
                if (down_done[k]) {                          // both ring slots of chunk k are free behind this point of the stream
                    hipError_t e = hipEventRecord(down_done[k], pc.down[t]);
                    if (e != hipSuccess) { sh.fail(e, "hipEventRecord (download stream)"); return; }
                    sh.mark(sh.down_issued, k);
                }
            }
            hipError_t e = hipStreamSynchronize(pc.down[t]);
            if (trace) fprintf(stderr, "[mf pipe] %8.2f ms  download thread %d drained\n", since(), t);
            if (e != hipSuccess) sh.fail(e, "hipStreamSynchronize (download stream)");
        });

    // the calling thread: the kernels of every chunk as soon as its frames have landed -- cell table + plan, warp, then (with `cropped`)
    // crop + resize
    int rc = MF_OK;
    hipError_t e = hipSuccess;
    for (int k = 0; k < nchunks && e == hipSuccess && rc == MF_OK; ++k) {
        if (!sh.wait(sh.up_ready, k)) break;
        const int i0 = k * chunk, i1 = chunk_end(k), m = i1 - i0;
        e = hipStreamWaitEvent(pc.compute, up_done[k], 0);
        if (e != hipSuccess) { sh.fail(e, "hipStreamWaitEvent"); break; }
        if (kernel_ms) (void)hipEventRecord(t0[k], pc.compute);
        if (job.warp) {
            const TableView tv = table_view(pc.table.p, m, W, H, R, C);
            rc = launch_cell_table(d_unstab + v2 * i0, d_stab + v2 * i0, m, W, H, R, C, tv, d_crop + 4 * (size_t)i0, (int32_t*)pc.status.p, pc.compute);
            if (rc == MF_OK) rc = launch_warp(slot_in(k), slot_out(k), tv, m, W, H, R, C, border, d_crop + 4 * (size_t)i0, pc.compute);
            if (rc != MF_OK) { sh.fail(hipErrorUnknown, "kernel launch"); break; }
            if (out) {
                e = hipEventRecord(warp_done[k], pc.compute);
                if (e != hipSuccess) { sh.fail(e, "hipEventRecord"); break; }
                sh.mark(sh.warp_ready, k);
            }
        }
        if (cropped) {                                     // _crop_frames (mfs.py:1111-1157) of this chunk, its source still in the caches
            rc = job.warp ? launch_crop_resize(slot_out(k), slot_in(k), m, W, H, rect[0], rect[1], rect[2], rect[3], pc.work.p, pc.compute)
                          : launch_crop_resize(slot_in(k), slot_out(k), m, W, H, rect[0], rect[1], rect[2], rect[3], pc.work.p, pc.compute);
            if (rc != MF_OK) { sh.fail(hipErrorUnknown, "kernel launch"); break; }
            e = hipEventRecord(resize_done[k], pc.compute);
            if (e != hipSuccess) { sh.fail(e, "hipEventRecord"); break; }
            sh.mark(sh.resize_ready, k);
        }
        if (kernel_ms) (void)hipEventRecord(t1[k], pc.compute);
        if (trace_level == 2) tr_k[k] = since();
    }
    if (!sh.abort && job.warp) {
        // per-frame crop values (and, without the early scan, the clip-level rectangle and the degenerate-mesh count) in one wait
        if (!cropped) rc = launch_crop_reduce(d_crop, n, W, H, (int32_t*)pc.bounds.p, pc.compute);
        if (rc != MF_OK) sh.fail(hipErrorUnknown, "kernel launch");
        else {
            e = hipMemcpyAsync(job.crop, d_crop, (size_t)n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, pc.compute);
            if (e == hipSuccess && !cropped) e = hipMemcpyAsync(rect, pc.bounds.p, sizeof rect, hipMemcpyDeviceToHost, pc.compute);
            if (e == hipSuccess && !cropped) e = hipMemcpyAsync(&status, pc.status.p, sizeof(int32_t), hipMemcpyDeviceToHost, pc.compute);
            if (e == hipSuccess) e = hipStreamSynchronize(pc.compute);
            if (e != hipSuccess) sh.fail(e, "download of the crop values");
        }
    }
    if (trace) fprintf(stderr, "[mf pipe] %8.2f ms  every chunk's kernels issued\n", since());
    for (auto& w : workers) w.join();
    if (trace) fprintf(stderr, "[mf pipe] %8.2f ms  done (%d frames of %d x %d, %d chunks of %d, %d ring slots)\n", since(), n, W, H, nchunks, chunk, slots);
    for (size_t k = 0; k < tr_up0.size(); ++k)
        fprintf(stderr, "[mf pipe] chunk %3zu  up %7.2f .. %7.2f   kernels issued %7.2f   down %7.2f .. %7.2f\n", k, tr_up0[k], tr_up1[k], tr_k[k], tr_dn0[k], tr_dn1[k]);
    if (job.bounds) for (int i = 0; i < 4; ++i) job.bounds[i] = rect[i];
    if (status != 0) {
        (void)hipDeviceSynchronize();
        set_error("%s: %d degenerate mesh cell(s): no homography exists (cv2.findHomography would return None)", job.name, status);
        return MF_ERR_DEGENERATE;
    }
    if (sh.abort) {
        (void)hipDeviceSynchronize();
        if (rc != MF_OK) return rc;                        // launch_* already set the message
        return hip_fail(sh.err, sh.what);
    }
    if (kernel_ms) {
        MF_HIP_TRY(hipStreamSynchronize(pc.compute));
        float total = 0.0f;
        for (int k = 0; k < nchunks; ++k) { float ms = 0.0f; MF_HIP_TRY(hipEventElapsedTime(&ms, t0[k], t1[k])); total += ms; }
        { float ms = 0.0f; MF_HIP_TRY(hipEventElapsedTime(&ms, r0, r1)); total += ms; }         // the rectangle pre-pass (tables + scan + reduce)
        *kernel_ms = total;
    }
    return MF_OK;
}

int warp_host_frames(const uint8_t* const* frames, uint8_t* const* out, uint8_t* const* cropped, const double* unstab, const double* stab,
                     int n, int W, int H, int R, int C, const uint8_t border_bgr[3], int32_t* crop, int32_t* bounds, float* kernel_ms)
{
    PipeJob job{};
    job.frames = frames; job.out = out; job.cropped = cropped; job.unstab = unstab; job.stab = stab;
    job.n = n; job.W = W; job.H = H; job.R = R; job.C = C;
    job.border = (uint32_t)border_bgr[0] | ((uint32_t)border_bgr[1] << 8) | ((uint32_t)border_bgr[2] << 16);
    job.crop = crop; job.bounds = bounds; job.kernel_ms = kernel_ms;
    job.warp = true;
    job.rect[0] = 0; job.rect[1] = 0; job.rect[2] = W - 1; job.rect[3] = H - 1;
    job.name = cropped ? "mf_warp_crop_u8c3_host_frames" : "mf_warp_u8c3_host";
    return run_host_pipeline(job);
}

}  // namespace mf

using namespace mf;

extern "C" {

static int check_host_args(const char* name, const uint8_t* const* frames, uint8_t* const* out, uint8_t* const* cropped, const double* unstab,
                           const double* stab, int n, int W, int H, int R, int C, const uint8_t* border_bgr, const int32_t* crop)
{
    if (!frames || !unstab || !stab || !border_bgr || !crop || (!out && !cropped)) { set_error("%s: null pointer", name); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || W < 2 || H < 2 || R <= 0 || C <= 0) { set_error("%s: bad sizes", name); return MF_ERR_INVALID_ARG; }
    for (int i = 0; i < n; ++i)
        if (!frames[i] || (out && !out[i]) || (cropped && !cropped[i])) { set_error("%s: null frame pointer %d", name, i); return MF_ERR_INVALID_ARG; }
    return MF_OK;
}

int mf_warp_u8c3_host_frames(const uint8_t* const* frames, uint8_t* const* out, const double* unstab, const double* stab, int n,
                             int W, int H, int R, int C, const uint8_t border_bgr[3], int32_t* crop, float* kernel_ms)
{
    if (!out) { set_error("mf_warp_u8c3_host_frames: null pointer"); return MF_ERR_INVALID_ARG; }
    if (const int rc = check_host_args("mf_warp_u8c3_host_frames", frames, out, nullptr, unstab, stab, n, W, H, R, C, border_bgr, crop)) return rc;
    return warp_host_frames(frames, out, nullptr, unstab, stab, n, W, H, R, C, border_bgr, crop, nullptr, kernel_ms);
}

int mf_warp_crop_u8c3_host_frames(const uint8_t* const* frames, uint8_t* const* out, uint8_t* const* cropped, const double* unstab,
                                  const double* stab, int n, int W, int H, int R, int C, const uint8_t border_bgr[3], int32_t* crop,
                                  int32_t bounds[4], float* kernel_ms)
{
    if (!cropped || !bounds) { set_error("mf_warp_crop_u8c3_host_frames: null pointer"); return MF_ERR_INVALID_ARG; }
    if (const int rc = check_host_args("mf_warp_crop_u8c3_host_frames", frames, out, cropped, unstab, stab, n, W, H, R, C, border_bgr, crop)) return rc;
    return warp_host_frames(frames, out, cropped, unstab, stab, n, W, H, R, C, border_bgr, crop, bounds, kernel_ms);
}

int mf_warp_u8c3_host(const uint8_t* frames, uint8_t* out, const double* unstab, const double* stab,
                      int n, int W, int H, int R, int C, const uint8_t border_bgr[3], int32_t* crop,
                      float* kernel_ms)
{
    if (!frames || !out || !unstab || !stab || !border_bgr || !crop) { set_error("mf_warp_u8c3_host: null pointer"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || W < 2 || H < 2 || R <= 0 || C <= 0) { set_error("mf_warp_u8c3_host: bad sizes"); return MF_ERR_INVALID_ARG; }
    const size_t fb = (size_t)W * H * 3;
    std::vector<const uint8_t*> in(n);
    std::vector<uint8_t*> outp(n);
    for (int i = 0; i < n; ++i) { in[i] = frames + fb * i; outp[i] = out + fb * i; }
    return warp_host_frames(in.data(), outp.data(), nullptr, unstab, stab, n, W, H, R, C, border_bgr, crop, nullptr, kernel_ms);
}

int mf_crop_resize_u8c3_host_frames(const uint8_t* const* frames, uint8_t* const* cropped, int n, int W, int H, int left, int top,
                                    int right, int bottom, float* kernel_ms)
{
    if (!frames || !cropped) { set_error("mf_crop_resize_u8c3_host_frames: null pointer"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || W < 1 || H < 1 || W > 32767 || H > 32767) { set_error("mf_crop_resize_u8c3_host_frames: unsupported shape n=%d W=%d H=%d", n, W, H); return MF_ERR_INVALID_ARG; }
    if (left < 0 || top < 0 || right >= W || bottom >= H || right < left || bottom < top) {        // before any output page is touched
        set_error("mf_crop_resize_u8c3_host_frames: empty or out-of-frame crop rectangle (%d, %d, %d, %d) for %dx%d (cv2.resize would "
                  "fail on an empty source)", left, top, right, bottom, W, H);
        return MF_ERR_INVALID_ARG;
    }
    for (int i = 0; i < n; ++i)
        if (!frames[i] || !cropped[i]) { set_error("mf_crop_resize_u8c3_host_frames: null frame pointer %d", i); return MF_ERR_INVALID_ARG; }
    PipeJob job{};
    job.frames = frames; job.cropped = cropped;
    job.n = n; job.W = W; job.H = H; job.R = 1; job.C = 1;
    job.kernel_ms = kernel_ms;
    job.warp = false;
    job.rect[0] = left; job.rect[1] = top; job.rect[2] = right; job.rect[3] = bottom;
    job.name = "mf_crop_resize_u8c3_host_frames";
    return run_host_pipeline(job);
}

int mf_host_cache_release(void)
{
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int d = 0; d < PIPE_MAX_DEVICES; ++d) {
        std::lock_guard<std::mutex> g(g_pipe[d].lock);
        if (g_pipe[d].device >= 0) {
            (void)hipSetDevice(g_pipe[d].device);
            (void)hipDeviceSynchronize();
            g_pipe[d].release();
        }
    }
    (void)hipSetDevice(prev);
    return MF_OK;
}

}  // extern "C"

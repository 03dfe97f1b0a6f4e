// What would the host pipeline gain from its OWN pinned staging?  A clip of `MiB` bytes goes up from pageable memory while another comes
// down into pageable memory (pre-faulted), in chunks of 50 MB on U + D host threads with a HIP stream each -- csrc/hostpipe.hip without
// the kernels:
//   direct   hipMemcpyAsync straight from / to the pageable buffers (the runtime stages them): what the pipeline does today
//   staged   memcpy into a pinned slot of this thread, hipMemcpyAsync pinned -> device (two slots per thread: the copy of chunk k + 1 is
//            prepared while chunk k's DMA runs); down: DMA into a pinned slot, then memcpy out
//   registr  hipHostRegister on each piece, DMA straight from it, hipHostUnregister (down as in `direct`)
// Up streams are PRIMED with a pageable copy first (tools/pcie_numa.hip: a stream whose first large copy was pageable host -> device runs
// its later pinned copies beside device -> host traffic at 48 GB/s per direction instead of 33).
// Build: make -C tools pcie_staged; run: tools/pcie_staged [MiB] [U] [D]
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        v.x ^= 3u;
        out[i] = v;
    }
}

int main(int argc, char** argv)
{
    // DEP=1: the `direct` mode with the pipeline's dependencies -- chunk k comes down only after it went up (event wait on the download
    // stream); KERNEL=1: ... and after a kernel on a compute stream has read it and written the chunk that comes down; RING=n: chunk k
    // goes up only after chunk k - n came down (event wait on the upload stream)
    const bool dep = getenv("DEP") || getenv("KERNEL") || getenv("RING");
    const bool with_kernel = getenv("KERNEL") != nullptr;
    const int ring = getenv("RING") ? atoi(getenv("RING")) : 0;
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 1780) << 20;
    const int U = argc > 2 ? atoi(argv[2]) : 4, D = argc > 3 ? atoi(argv[3]) : 4;
    const size_t chunk = getenv("CHUNK_KB") ? (size_t)atol(getenv("CHUNK_KB")) << 10 : (size_t)(getenv("CHUNK_MB") ? atol(getenv("CHUNK_MB")) : 50) << 20;
    // FRAME_KB / FRAME_DN_KB: the direct mode's copies cut into pieces of that size (frames that are separate allocations: 6,075 KB at 1080p)
    const size_t piece = getenv("FRAME_KB") ? (size_t)atol(getenv("FRAME_KB")) << 10 : chunk;
    const size_t piece_dn = getenv("FRAME_DN_KB") ? (size_t)atol(getenv("FRAME_DN_KB")) << 10 : chunk;
    if (U == 0 || D == 0) fprintf(stderr, "one direction only\n");
    const int nchunks = (int)((bytes + chunk - 1) / chunk);
    uint8_t *d_up, *d_dn;
    CK(hipMalloc(&d_up, bytes));
    CK(hipMalloc(&d_dn, bytes));
    CK(hipMemset(d_dn, 2, bytes));
    // FRESH=p: the output buffer is a NEW anonymous mapping at every repetition (np.empty of the caller) and p threads fault its pages in
    // chunk by chunk ahead of the downloads, as the pipeline's populate threads do; THP=1|2|3: madvise(MADV_HUGEPAGE) on the output (1),
    // the input (2) or both (3)
    const int fresh = getenv("FRESH") ? atoi(getenv("FRESH")) : 0;
    const int thp = getenv("THP") ? atoi(getenv("THP")) : 0;
    auto map_anon = [&](bool huge) {
        void* p = mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) { perror("mmap"); exit(1); }
        uint8_t* a = (uint8_t*)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
        if (huge && madvise(a, bytes, MADV_HUGEPAGE) != 0) perror("madvise(MADV_HUGEPAGE)");
        return std::make_pair(p, a);
    };
    // OFF_UP / OFF_DN: bytes added to the host addresses (glibc hands large arrays out at page + 16)
    const size_t off_up = getenv("OFF_UP") ? atol(getenv("OFF_UP")) : 0, off_dn = getenv("OFF_DN") ? atol(getenv("OFF_DN")) : 0;
    uint8_t* h_up = map_anon(thp & 2).second + off_up;
    auto dn_map = map_anon(thp & 1);
    uint8_t* h_dn = dn_map.second + off_dn;
    memset(h_up, 1, bytes);
    memset(h_dn, 0, bytes);
    std::vector<hipStream_t> su(U), sd(D);
    for (auto& s : su) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& s : sd) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (!getenv("NO_PRIME")) {
        for (int t = 0; t < U; ++t) CK(hipMemcpyAsync(d_up, h_up + (size_t)t * (16 << 20), 16 << 20, hipMemcpyHostToDevice, su[t]));
        CK(hipDeviceSynchronize());
    }
    std::vector<uint8_t*> pin_up(2 * U), pin_dn(2 * D);
    for (auto& p : pin_up) CK(hipHostMalloc(&p, chunk, hipHostMallocDefault));
    for (auto& p : pin_dn) CK(hipHostMalloc(&p, chunk, hipHostMallocDefault));
    for (auto p : pin_up) memset(p, 0, chunk);
    for (auto p : pin_dn) memset(p, 0, chunk);
    auto size_of = [&](int k) { return (size_t)k * chunk + chunk <= bytes ? chunk : bytes - (size_t)k * chunk; };
    hipStream_t compute;
    CK(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
    std::vector<hipEvent_t> up_done(nchunks), warp_done(nchunks), down_done(nchunks);
    for (auto* v : {&up_done, &warp_done, &down_done}) for (auto& e : *v) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<std::atomic<int>> up_ready(nchunks), warp_ready(nchunks), down_issued(nchunks), populated(nchunks);
    auto wait_flag = [](std::atomic<int>& f) { while (!f.load(std::memory_order_acquire)) std::this_thread::yield(); };
    // RECREATE=n: the `direct` mode n times, each time on a NEW set of streams (the old ones destroyed first): is the duplex rate a
    // property of the process or of the streams it happens to hold?
    const int recreate = getenv("RECREATE") ? atoi(getenv("RECREATE")) : 0;
    for (int mode = 0, round = 0; mode < 3; ++mode) {
        if (recreate && mode == 1) {
            if (++round >= recreate) break;
            mode = 0;
            for (auto& s_ : su) { CK(hipStreamDestroy(s_)); }
            for (auto& s_ : sd) { CK(hipStreamDestroy(s_)); }
            if (getenv("RECREATE_EXTRA")) {                  // shift the queue assignment: a few streams that stay alive
                hipStream_t extra;
                for (int i = 0; i < atoi(getenv("RECREATE_EXTRA")); ++i) { CK(hipStreamCreateWithFlags(&extra, hipStreamNonBlocking)); CK(hipMemsetAsync(d_up, 0, 64, extra)); }
            }
            for (auto& s_ : su) CK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
            for (auto& s_ : sd) CK(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
            if (!getenv("NO_PRIME")) {
                for (int t = 0; t < U; ++t) CK(hipMemcpyAsync(d_up, h_up + (size_t)t * (16 << 20), 16 << 20, hipMemcpyHostToDevice, su[t]));
                CK(hipDeviceSynchronize());
            }
        }
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            for (int k = 0; k < nchunks; ++k) { up_ready[k] = 0; warp_ready[k] = 0; down_issued[k] = 0; populated[k] = fresh ? 0 : 1; }
            if (fresh) {
                munmap(dn_map.first, bytes + (2 << 20));
                dn_map = map_anon(thp & 1);
                h_dn = dn_map.second + off_dn;
            }
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < fresh; ++t)
                th.emplace_back([&, t] {
                    for (int k = t; k < nchunks; k += fresh) {
                        uint8_t* a = h_dn + (size_t)k * chunk;
                        for (size_t o = 0; o < size_of(k); o += 4096) *(volatile uint8_t*)(a + o) = 0;
                        populated[k].store(1, std::memory_order_release);
                    }
                });
            if (mode == 0 && dep)
                th.emplace_back([&] {                     // the calling thread of the pipeline: the kernels of chunk k behind its upload
                    for (int k = 0; k < nchunks; ++k) {
                        wait_flag(up_ready[k]);
                        CK(hipStreamWaitEvent(compute, up_done[k], 0));
                        if (with_kernel)
                            touch_kernel<<<1024, 256, 0, compute>>>((const uint4*)(d_up + (size_t)k * chunk), (uint4*)(d_dn + (size_t)k * chunk), size_of(k) / 16);
                        CK(hipEventRecord(warp_done[k], compute));
                        warp_ready[k].store(1, std::memory_order_release);
                    }
                });
            for (int t = 0; t < U; ++t)
                th.emplace_back([&, t] {
                    hipEvent_t ev[2];
                    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    int used = 0;
                    for (int k = t; k < nchunks; k += U, ++used) {
                        const size_t off = (size_t)k * chunk, m = size_of(k);
                        if (mode == 0) {
                            if (ring && k >= ring) { wait_flag(down_issued[k - ring]); CK(hipStreamWaitEvent(su[t], down_done[k - ring], 0)); }
                            for (size_t o = 0; o < m; o += piece) CK(hipMemcpyAsync(d_up + off + o, h_up + off + o, m - o < piece ? m - o : piece, hipMemcpyHostToDevice, su[t]));
                            if (dep) { CK(hipEventRecord(up_done[k], su[t])); up_ready[k].store(1, std::memory_order_release); }
                            continue;
                        }
                        if (mode == 2) {        // register each piece in place, DMA from it, unregister when the chunk has gone up
                            for (size_t o = 0; o < m; o += piece) {
                                const size_t n = m - o < piece ? m - o : piece;
                                CK(hipHostRegister(h_up + off + o, n, hipHostRegisterDefault));
                                CK(hipMemcpyAsync(d_up + off + o, h_up + off + o, n, hipMemcpyHostToDevice, su[t]));
                            }
                            CK(hipStreamSynchronize(su[t]));
                            for (size_t o = 0; o < m; o += piece) CK(hipHostUnregister(h_up + off + o));
                            continue;
                        }
                        uint8_t* slot = pin_up[2 * t + (used & 1)];
                        if (used >= 2) CK(hipEventSynchronize(ev[used & 1]));       // the DMA that last read this slot
                        memcpy(slot, h_up + off, m);
                        CK(hipMemcpyAsync(d_up + off, slot, m, hipMemcpyHostToDevice, su[t]));
                        CK(hipEventRecord(ev[used & 1], su[t]));
                    }
                    CK(hipStreamSynchronize(su[t]));
                    for (auto& e : ev) CK(hipEventDestroy(e));
                });
            for (int t = 0; t < D; ++t)
                th.emplace_back([&, t] {
                    int used = 0, prev_k = -1;
                    hipEvent_t ev[2];
                    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    for (int k = t; k < nchunks; k += D, ++used) {
                        const size_t off = (size_t)k * chunk, m = size_of(k);
                        if (mode != 1) {
                            wait_flag(populated[k]);
                            if (mode == 0 && dep) { wait_flag(warp_ready[k]); CK(hipStreamWaitEvent(sd[t], warp_done[k], 0)); }
                            for (size_t o = 0; o < m; o += piece_dn) CK(hipMemcpyAsync(h_dn + off + o, d_dn + off + o, m - o < piece_dn ? m - o : piece_dn, hipMemcpyDeviceToHost, sd[t]));
                            if (mode == 0 && dep) { CK(hipEventRecord(down_done[k], sd[t])); down_issued[k].store(1, std::memory_order_release); }
                            continue;
                        }
                        CK(hipMemcpyAsync(pin_dn[2 * t + (used & 1)], d_dn + off, m, hipMemcpyDeviceToHost, sd[t]));      // DMA of chunk k ...
                        CK(hipEventRecord(ev[used & 1], sd[t]));
                        if (prev_k >= 0) {                                                                         // ... while chunk k - D is copied out
                            CK(hipEventSynchronize(ev[(used - 1) & 1]));
                            memcpy(h_dn + (size_t)prev_k * chunk, pin_dn[2 * t + ((used - 1) & 1)], size_of(prev_k));
                        }
                        prev_k = k;
                    }
                    if (mode == 1 && prev_k >= 0) {
                        CK(hipEventSynchronize(ev[(used - 1) & 1]));
                        memcpy(h_dn + (size_t)prev_k * chunk, pin_dn[2 * t + ((used - 1) & 1)], size_of(prev_k));
                    }
                    CK(hipStreamSynchronize(sd[t]));
                    for (auto& e : ev) CK(hipEventDestroy(e));
                });
            for (auto& t : th) t.join();
            const double dt = now() - t0;
            if (rep > 0 && dt < best) best = dt;
        }
        printf("%-7s: %6.1f ms for %zu MiB each way = %5.1f GB/s per direction (%d + %d threads)%s\n", mode == 2 ? "registr" : mode ? "staged" : "direct", best * 1e3, bytes >> 20,
               bytes / best / 1e9, U, D, h_dn[bytes - 1] == 2 ? "" : "  [WRONG DATA]");
        memset(h_dn, 0, bytes);
    }
    return 0;
}

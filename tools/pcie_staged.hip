// What would the host pipeline gain from its OWN pinned staging?  A clip of `MiB` bytes goes up from pageable memory while another comes
// down into pageable memory (pre-faulted), in chunks of 50 MB on U + D host threads with a HIP stream each -- csrc/hostpipe.hip without
// the kernels:
//   direct   hipMemcpyAsync straight from / to the pageable buffers (the runtime stages them): what the pipeline does today
//   staged   memcpy into a pinned slot of this thread, hipMemcpyAsync pinned -> device (two slots per thread: the copy of chunk k + 1 is
//            prepared while chunk k's DMA runs); down: DMA into a pinned slot, then memcpy out
// Up streams are PRIMED with a pageable copy first (tools/pcie_numa.hip: a stream whose first large copy was pageable host -> device runs
// its later pinned copies beside device -> host traffic at 48 GB/s per direction instead of 33).
// Build: make -C tools pcie_staged; run: tools/pcie_staged [MiB] [U] [D]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 1780) << 20;
    const int U = argc > 2 ? atoi(argv[2]) : 4, D = argc > 3 ? atoi(argv[3]) : 4;
    const size_t chunk = (size_t)(getenv("CHUNK_MB") ? atol(getenv("CHUNK_MB")) : 50) << 20;
    // FRAME_KB / FRAME_DN_KB: the direct mode's copies cut into pieces of that size (frames that are separate allocations: 6,075 KB at 1080p)
    const size_t piece = getenv("FRAME_KB") ? (size_t)atol(getenv("FRAME_KB")) << 10 : chunk;
    const size_t piece_dn = getenv("FRAME_DN_KB") ? (size_t)atol(getenv("FRAME_DN_KB")) << 10 : chunk;
    if (U == 0 || D == 0) fprintf(stderr, "one direction only\n");
    const int nchunks = (int)((bytes + chunk - 1) / chunk);
    uint8_t *d_up, *d_dn;
    CK(hipMalloc(&d_up, bytes));
    CK(hipMalloc(&d_dn, bytes));
    CK(hipMemset(d_dn, 2, bytes));
    uint8_t* h_up = (uint8_t*)aligned_alloc(4096, bytes);
    uint8_t* h_dn = (uint8_t*)aligned_alloc(4096, bytes);
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
    for (int mode = 0; mode < 2; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < U; ++t)
                th.emplace_back([&, t] {
                    hipEvent_t ev[2];
                    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    int used = 0;
                    for (int k = t; k < nchunks; k += U, ++used) {
                        const size_t off = (size_t)k * chunk, m = size_of(k);
                        if (mode == 0) {
                            for (size_t o = 0; o < m; o += piece) CK(hipMemcpyAsync(d_up + off + o, h_up + off + o, m - o < piece ? m - o : piece, hipMemcpyHostToDevice, su[t]));
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
                        if (mode == 0) {
                            for (size_t o = 0; o < m; o += piece_dn) CK(hipMemcpyAsync(h_dn + off + o, d_dn + off + o, m - o < piece_dn ? m - o : piece_dn, hipMemcpyDeviceToHost, sd[t]));
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
        printf("%-7s: %6.1f ms for %zu MiB each way = %5.1f GB/s per direction (%d + %d threads)%s\n", mode ? "staged" : "direct", best * 1e3, bytes >> 20,
               bytes / best / 1e9, U, D, h_dn[bytes - 1] == 2 ? "" : "  [WRONG DATA]");
        memset(h_dn, 0, bytes);
    }
    return 0;
}

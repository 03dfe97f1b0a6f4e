// PCIe duplex rate by NUMA placement of the HOST buffers: for every NUMA node of the box, pin this process to the node's CPUs, allocate
// + first-touch pageable and pinned buffers there, and time 1 GiB up while 1 GiB comes down (hipMemcpyAsync, 64 MiB pieces over four
// streams per direction -- how csrc/hostpipe.hip moves a clip).  Build: make -C tools pcie_numa; run: tools/pcie_numa [MiB]
#include <hip/hip_runtime.h>
#include <sched.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool cpus_of_node(int node, cpu_set_t* set)
{
    char path[128];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096] = "";
    if (!fgets(buf, sizeof buf, f)) { fclose(f); return false; }
    fclose(f);
    CPU_ZERO(set);
    for (char* tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a, b;
        if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; ++c) CPU_SET(c, set); }
        else if (sscanf(tok, "%d", &a) == 1) CPU_SET(a, set);
    }
    return CPU_COUNT(set) > 0;
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 20;
    const size_t piece = 64u << 20;
    uint8_t *d_up, *d_dn;
    CK(hipMalloc(&d_up, bytes));
    CK(hipMalloc(&d_dn, bytes));
    CK(hipMemset(d_dn, 2, bytes));
    hipStream_t su[4], sd[4];
    for (auto& s : su) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& s : sd) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // PRIME=tiny: one 4 KB PAGEABLE copy each way on every stream before anything else (does the first use of a stream decide which
    // engines its copies take?); PRIME=other: the same on OTHER streams, destroyed afterwards
    if (const char* pb = getenv("PRIME_BIG")) {           // "u", "d", "ud" (one after the other), "x" (both at once): 64 MiB PAGEABLE copies on stream 0 of each direction
        uint8_t* pu = (uint8_t*)aligned_alloc(4096, piece);
        uint8_t* pd = (uint8_t*)aligned_alloc(4096, piece);
        memset(pu, 1, piece); memset(pd, 0, piece);
        for (const char* c = pb; *c; ++c) {
            if (*c == 'u' || *c == 'x') CK(hipMemcpyAsync(d_up, pu, piece, hipMemcpyHostToDevice, su[0]));
            if (*c == 'd' || *c == 'x') CK(hipMemcpyAsync(pd, d_dn, piece, hipMemcpyDeviceToHost, sd[0]));
            CK(hipDeviceSynchronize());
        }
        printf("primed with 64 MiB pageable copies: %s\n", pb);
        free(pu); free(pd);
    }
    if (const char* pa = getenv("PRIME_ALL")) {           // N MiB of PAGEABLE host -> device on EACH of the four up streams, at once
        const size_t nb = (size_t)atol(pa) << 20;
        uint8_t* pu;
        const bool prime_pinned = getenv("PRIME_PINNED") != nullptr;          // (experiment) the priming copies from PINNED memory instead
        if (prime_pinned) CK(hipHostMalloc(&pu, 4 * nb, hipHostMallocDefault)); else pu = (uint8_t*)aligned_alloc(4096, 4 * nb);
        memset(pu, 1, 4 * nb);
        for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(d_up + k * nb, pu + k * nb, nb, hipMemcpyHostToDevice, su[k]));
        CK(hipDeviceSynchronize());
        printf("primed with %zu MiB %s host -> device copies on each up stream\n", nb >> 20, prime_pinned ? "PINNED" : "pageable");
        if (prime_pinned) CK(hipHostFree(pu)); else free(pu);
    }
    if (const char* pr = getenv("PRIME")) {
        static uint8_t small_up[4096], small_dn[4096];
        hipStream_t other[8];
        const bool on_other = strcmp(pr, "other") == 0;
        if (on_other) for (auto& s : other) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int k = 0; k < 4; ++k) {
            CK(hipMemcpyAsync(d_up, small_up, sizeof small_up, hipMemcpyHostToDevice, on_other ? other[k] : su[k]));
            CK(hipMemcpyAsync(small_dn, d_dn, sizeof small_dn, hipMemcpyDeviceToHost, on_other ? other[4 + k] : sd[k]));
        }
        CK(hipDeviceSynchronize());
        if (on_other) for (auto& s : other) CK(hipStreamDestroy(s));
        printf("primed with 4 KB pageable copies on %s streams\n", on_other ? "other" : "the same");
    }
    cpu_set_t all;
    sched_getaffinity(0, sizeof all, &all);
    printf("%zu MiB each way; this process may use %d CPUs\n", bytes >> 20, CPU_COUNT(&all));
    const int last_node = argc > 3 ? atoi(argv[3]) : 16;
    for (int node = -1; node < last_node; ++node) {
        cpu_set_t set;
        if (node >= 0) {
            if (!cpus_of_node(node, &set)) break;
            cpu_set_t both;
            CPU_AND(&both, &set, &all);
            if (CPU_COUNT(&both) == 0) { printf("node %d: none of its CPUs allowed here\n", node); continue; }
            if (sched_setaffinity(0, sizeof both, &both) != 0) { printf("node %d: sched_setaffinity failed\n", node); continue; }
        }
        const char* order = argc > 2 ? argv[2] : "pP";      // p = pageable buffers, P = pinned buffers, in this order
        for (const char* o = order; *o; ++o) {
            const int pinned = *o == 'P';
            uint8_t *h_up, *h_dn;
            if (pinned) { CK(hipHostMalloc(&h_up, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&h_dn, bytes, hipHostMallocDefault)); }
            else { h_up = (uint8_t*)aligned_alloc(4096, bytes); h_dn = (uint8_t*)aligned_alloc(4096, bytes); }
            memset(h_up, 1, bytes);          // first touch on this node
            memset(h_dn, 0, bytes);
            double best[3] = { 1e9, 1e9, 1e9 };
            const int pmodes = getenv("P_MODES") ? atoi(getenv("P_MODES")) : 7;      // (experiment) which modes the PAGEABLE phase runs: bit 0 up, 1 down, 2 both
            const int preps = getenv("P_REPS") ? atoi(getenv("P_REPS")) : 4;
            for (int mode = 0; mode < 3; ++mode)        // 0: up alone, 1: down alone, 2: both
                for (int rep = 0; rep < (pinned ? 4 : preps); ++rep) {
                    if (!pinned && !((pmodes >> mode) & 1)) continue;
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    int k = 0;
                    for (size_t off = 0; off < bytes; off += piece, ++k) {
                        const size_t m = bytes - off < piece ? bytes - off : piece;
                        if (mode != 1) CK(hipMemcpyAsync(d_up + off, h_up + off, m, hipMemcpyHostToDevice, su[k & 3]));
                        if (mode != 0) CK(hipMemcpyAsync(h_dn + off, d_dn + off, m, hipMemcpyDeviceToHost, sd[k & 3]));
                    }
                    CK(hipDeviceSynchronize());
                    const double dt = now() - t0;
                    if (rep > 0 && dt < best[mode]) best[mode] = dt;
                }
            printf("node %2d %-8s: up alone %5.1f  down alone %5.1f  both %5.1f GB/s per direction\n", node, pinned ? "pinned" : "pageable",
                   bytes / best[0] / 1e9, bytes / best[1] / 1e9, bytes / best[2] / 1e9);
            fflush(stdout);
            if (pinned) { CK(hipHostFree(h_up)); CK(hipHostFree(h_dn)); } else if (!getenv("KEEP")) { free(h_up); free(h_dn); }
        }
        if (node >= 0) sched_setaffinity(0, sizeof all, &all);
    }
    return 0;
}

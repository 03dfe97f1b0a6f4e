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
    cpu_set_t all;
    sched_getaffinity(0, sizeof all, &all);
    printf("%zu MiB each way; this process may use %d CPUs\n", bytes >> 20, CPU_COUNT(&all));
    for (int node = -1; node < 16; ++node) {
        cpu_set_t set;
        if (node >= 0) {
            if (!cpus_of_node(node, &set)) break;
            cpu_set_t both;
            CPU_AND(&both, &set, &all);
            if (CPU_COUNT(&both) == 0) { printf("node %d: none of its CPUs allowed here\n", node); continue; }
            if (sched_setaffinity(0, sizeof both, &both) != 0) { printf("node %d: sched_setaffinity failed\n", node); continue; }
        }
        for (int pinned = 0; pinned < 2; ++pinned) {
            uint8_t *h_up, *h_dn;
            if (pinned) { CK(hipHostMalloc(&h_up, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&h_dn, bytes, hipHostMallocDefault)); }
            else { h_up = (uint8_t*)aligned_alloc(4096, bytes); h_dn = (uint8_t*)aligned_alloc(4096, bytes); }
            memset(h_up, 1, bytes);          // first touch on this node
            memset(h_dn, 0, bytes);
            double best[3] = { 1e9, 1e9, 1e9 };
            for (int mode = 0; mode < 3; ++mode)        // 0: up alone, 1: down alone, 2: both
                for (int rep = 0; rep < 4; ++rep) {
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
            if (pinned) { CK(hipHostFree(h_up)); CK(hipHostFree(h_dn)); } else { free(h_up); free(h_dn); }
        }
        if (node >= 0) sched_setaffinity(0, sizeof all, &all);
    }
    return 0;
}

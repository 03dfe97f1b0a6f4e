// Multi-GPU exchange steps of the path in the C ABI, on RCCL directly: one process drives all the
// GPUs of a node (ncclCommInitAll), no torch.
//
//   mf_allreduce_crop   clip-level crop rectangle (mfs.py:1103-1106): {max left, max top, min right, min bottom} over the
//                       ranks' shard-level rectangles -- 16 bytes, one grouped RCCL call.
//   mf_gather_frames    north_star's final gather: every rank's stabilized frame shard to one GPU over xGMI.  Shards may be
//                       ragged (300 frames / 8 GPUs), so it is one group of ncclSend / ncclRecv with per-rank byte counts
//                       instead of ncclGather (which needs equal counts, rccl.h:745).
//
// librccl is opened at mf_comm_init_all() with dlopen (no link-time dependency: a process that already holds an RCCL --
// e.g. one that imported torch -- shares that copy; a process without RCCL can still use every single-GPU entry point).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "mf_common.h"

namespace mf {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
    int ndev = 0;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};

Rccl g_rccl;
Comm g_comm;
std::mutex g_comm_lock;          // the communicator is process-wide state: every entry point below holds this

// Frees what a Comm holds (streams, communicators), on the right devices; keeps the caller's current device.
int destroy_comm(Comm& c)
{
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = MF_OK;
    for (int g = 0; g < c.ndev; ++g) {
        (void)hipSetDevice(g);
        if (g < (int)c.streams.size() && c.streams[g]) (void)hipStreamDestroy(c.streams[g]);
        if (g < (int)c.comms.size() && c.comms[g] && g_rccl.CommDestroy) {
            const ncclResult_t r = g_rccl.CommDestroy(c.comms[g]);
            if (r != ncclSuccess && rc == MF_OK) rc = -1000 - (int)r;
        }
    }
    (void)hipSetDevice(prev);
    c = Comm();
    return rc;
}

int nccl_fail(ncclResult_t r, const char* what)
{
    if (r == ncclSuccess) return MF_OK;
    set_error("%s: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error", (int)r);
    return -1000 - (int)r;                                  // -1000 - ncclResult_t
}

#define MF_NCCL_TRY(expr)                                  \
    do {                                                   \
        int _rc = nccl_fail((expr), #expr);                \
        if (_rc != MF_OK) return _rc;                      \
    } while (0)

int load_rccl()
{
    if (g_rccl.handle) return MF_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { set_error("mf_comm_init_all: cannot open librccl (%s)", dlerror()); return MF_ERR_HIP; }
    Rccl r;
    r.handle = h;
#define SYM(field, name) \
    *(void**)(&r.field) = dlsym(h, name); \
    if (!r.field) { set_error("mf_comm_init_all: librccl has no %s", name); dlclose(h); return MF_ERR_HIP; }
    SYM(CommInitAll, "ncclCommInitAll") SYM(CommDestroy, "ncclCommDestroy") SYM(AllReduce, "ncclAllReduce")
    SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = r;
    return MF_OK;
}

int sync_all()
{
    int prev = 0;
    MF_HIP_TRY(hipGetDevice(&prev));
    for (int g = 0; g < g_comm.ndev; ++g) {
        MF_HIP_TRY(hipSetDevice(g));
        MF_HIP_TRY(hipStreamSynchronize(g_comm.streams[g]));
    }
    MF_HIP_TRY(hipSetDevice(prev));
    return MF_OK;
}

}  // namespace
}  // namespace mf

using namespace mf;

extern "C" {

int mf_comm_init_all(int ndev)
{
    std::lock_guard<std::mutex> guard(g_comm_lock);
    if (g_comm.ndev != 0) { set_error("mf_comm_init_all: a communicator already exists (mf_comm_destroy first)"); return MF_ERR_INVALID_ARG; }
    int count = 0;
    MF_HIP_TRY(hipGetDeviceCount(&count));
    if (ndev <= 0 || ndev > count) { set_error("mf_comm_init_all: ndev=%d, %d device(s) visible", ndev, count); return MF_ERR_INVALID_ARG; }
    int rc = load_rccl();
    if (rc != MF_OK) return rc;
    int prev = 0;
    MF_HIP_TRY(hipGetDevice(&prev));
    Comm c;
    c.ndev = ndev;
    c.comms.assign(ndev, nullptr);
    c.streams.assign(ndev, nullptr);
    std::vector<int> devs(ndev);
    for (int g = 0; g < ndev; ++g) devs[g] = g;
    MF_NCCL_TRY(g_rccl.CommInitAll(c.comms.data(), ndev, devs.data()));
    hipError_t e = hipSuccess;
    for (int g = 0; g < ndev && e == hipSuccess; ++g) {
        e = hipSetDevice(g);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c.streams[g], hipStreamNonBlocking);
    }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {                         // part-way failure: give the communicators and streams back
        (void)destroy_comm(c);
        return hip_fail(e, "mf_comm_init_all: stream creation");
    }
    g_comm = c;
    return MF_OK;
}

int mf_comm_size(int* ndev)
{
    if (!ndev) { set_error("mf_comm_size: null"); return MF_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> guard(g_comm_lock);
    *ndev = g_comm.ndev;
    return MF_OK;
}

int mf_comm_destroy(void)
{
    std::lock_guard<std::mutex> guard(g_comm_lock);
    if (g_comm.ndev == 0) return MF_OK;
    const int rc = destroy_comm(g_comm);
    if (rc != MF_OK) set_error("mf_comm_destroy: ncclCommDestroy failed (%d)", -1000 - rc);
    return rc;
}

// An RCCL call that fails between ncclGroupStart and ncclGroupEnd must not leave the group open (every later RCCL call of the
// process would join it): close it, then report the first error.
#define MF_NCCL_IN_GROUP(expr)                             \
    do {                                                   \
        if (rc == MF_OK) rc = nccl_fail((expr), #expr);    \
    } while (0)

int mf_allreduce_crop(int32_t* const* d_bounds)
{
    std::lock_guard<std::mutex> guard(g_comm_lock);
    if (g_comm.ndev == 0) { set_error("mf_allreduce_crop: no communicator (mf_comm_init_all first)"); return MF_ERR_INVALID_ARG; }
    if (!d_bounds) { set_error("mf_allreduce_crop: null"); return MF_ERR_INVALID_ARG; }
    for (int g = 0; g < g_comm.ndev; ++g)
        if (!d_bounds[g]) { set_error("mf_allreduce_crop: null pointer for device %d", g); return MF_ERR_INVALID_ARG; }
    // {left, top} take the maximum, {right, bottom} the minimum (mfs.py:1103-1106): two 8-byte reductions in ONE group
    MF_NCCL_TRY(g_rccl.GroupStart());
    int rc = MF_OK;
    for (int g = 0; g < g_comm.ndev; ++g) {
        MF_NCCL_IN_GROUP(g_rccl.AllReduce(d_bounds[g], d_bounds[g], 2, ncclInt32, ncclMax, g_comm.comms[g], g_comm.streams[g]));
        MF_NCCL_IN_GROUP(g_rccl.AllReduce(d_bounds[g] + 2, d_bounds[g] + 2, 2, ncclInt32, ncclMin, g_comm.comms[g], g_comm.streams[g]));
    }
    const ncclResult_t end = g_rccl.GroupEnd();            // always: closes the group even after a failed call
    if (rc != MF_OK) return rc;
    MF_NCCL_TRY(end);
    return sync_all();
}

int mf_gather_frames(const uint8_t* const* d_shards, const size_t* shard_bytes, uint8_t* d_dst, int root)
{
    std::lock_guard<std::mutex> guard(g_comm_lock);
    if (g_comm.ndev == 0) { set_error("mf_gather_frames: no communicator (mf_comm_init_all first)"); return MF_ERR_INVALID_ARG; }
    if (!d_shards || !shard_bytes || !d_dst || root < 0 || root >= g_comm.ndev) { set_error("mf_gather_frames: bad arguments"); return MF_ERR_INVALID_ARG; }
    for (int g = 0; g < g_comm.ndev; ++g)                  // argument errors before the group opens
        if (shard_bytes[g] != 0 && !d_shards[g]) { set_error("mf_gather_frames: null shard for device %d", g); return MF_ERR_INVALID_ARG; }
    MF_NCCL_TRY(g_rccl.GroupStart());
    int rc = MF_OK;
    size_t offset = 0;
    for (int g = 0; g < g_comm.ndev; ++g) {
        if (shard_bytes[g] == 0) continue;
        MF_NCCL_IN_GROUP(g_rccl.Send(d_shards[g], shard_bytes[g], ncclUint8, root, g_comm.comms[g], g_comm.streams[g]));
        MF_NCCL_IN_GROUP(g_rccl.Recv(d_dst + offset, shard_bytes[g], ncclUint8, g, g_comm.comms[root], g_comm.streams[root]));
        offset += shard_bytes[g];
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (rc != MF_OK) return rc;
    MF_NCCL_TRY(end);
    return sync_all();
}

}  // extern "C"

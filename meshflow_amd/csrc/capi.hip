// C ABI of libmeshflow_hip.so: argument checking, error reporting, device plumbing and the
// host-buffer convenience wrappers.  Declarations and reference citations: include/meshflow_hip.h.
#include <stdarg.h>
#include <stdio.h>

#include "mf_common.h"

namespace mf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what)
{
    if (e == hipSuccess) return MF_OK;
    set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return MF_ERR_HIP;
}

static uint32_t pack_border(const uint8_t b[3]) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16); }

}  // namespace mf

using namespace mf;

extern "C" {

int mf_abi_version(void) { return MF_ABI_VERSION; }
const char* mf_last_error(void) { return g_err; }

int mf_device_count(int* count)
{
    if (!count) { set_error("mf_device_count: null"); return MF_ERR_INVALID_ARG; }
    *count = 0;
    MF_HIP_TRY(hipGetDeviceCount(count));
    return MF_OK;
}

int mf_set_device(int device)
{
    MF_HIP_TRY(hipSetDevice(device));
    return check_d16_zero_fill(nullptr);       // the one-time, synchronising device check of the byte-tap kernels: here, not in a launch
}

int mf_malloc(void** d_ptr, size_t bytes)
{
    if (!d_ptr) { set_error("mf_malloc: null"); return MF_ERR_INVALID_ARG; }
    MF_HIP_TRY(hipMalloc(d_ptr, bytes));
    return MF_OK;
}
int mf_free(void* d_ptr) { MF_HIP_TRY(hipFree(d_ptr)); return MF_OK; }
int mf_malloc_host(void** h_ptr, size_t bytes)
{
    if (!h_ptr) { set_error("mf_malloc_host: null"); return MF_ERR_INVALID_ARG; }
    MF_HIP_TRY(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return MF_OK;
}
int mf_free_host(void* h_ptr) { MF_HIP_TRY(hipHostFree(h_ptr)); return MF_OK; }
int mf_memcpy_h2d(void* d, const void* h, size_t bytes, void* stream)
{
    MF_HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return MF_OK;
}
int mf_memcpy_d2h(void* h, const void* d, size_t bytes, void* stream)
{
    MF_HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return MF_OK;
}
int mf_stream_synchronize(void* stream) { MF_HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return MF_OK; }

int mf_jacobi_f64(const double* d_b, double* d_x, const double* d_taps, const double* d_lam,
                  const double* d_inv_on, int F, int S, int omega, int iters, void* stream)
{
    if (!d_b || !d_x || !d_taps || !d_lam || !d_inv_on) { set_error("mf_jacobi_f64: null pointer"); return MF_ERR_INVALID_ARG; }
    if (d_b == d_x) { set_error("mf_jacobi_f64: d_b and d_x alias"); return MF_ERR_INVALID_ARG; }
    return launch_jacobi(d_b, d_x, d_taps, d_lam, d_inv_on, F, S, omega, iters, (hipStream_t)stream);
}

size_t mf_cell_table_bytes(int n, int W, int H, int R, int C)
{
    if (n <= 0 || R <= 0 || C <= 0 || W <= 0 || H <= 0) return 0;
    return table_bytes(n, W, H, R, C);
}

size_t mf_cell_table_bounds_offset(int n, int W, int H, int R, int C)
{
    if (n <= 0 || R <= 0 || C <= 0 || W <= 0 || H <= 0) return 0;
    alignas(16) static char origin[16];                    // (any address: only the offset of the section is wanted)
    const TableView tv = table_view(origin, n, W, H, R, C);
    return (size_t)((const char*)tv.bounds - origin);
}

int mf_cell_table_f64(const double* d_unstab, const double* d_stab, int n, int W, int H, int R, int C,
                      void* d_table, int32_t* d_crop, int32_t* d_status, void* stream)
{
    if (!d_unstab || !d_stab || !d_table || !d_crop || !d_status) { set_error("mf_cell_table_f64: null pointer"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0) { set_error("mf_cell_table_f64: bad sizes"); return MF_ERR_INVALID_ARG; }
    const TableView tv = table_view(d_table, n, W, H, R, C);
    return launch_cell_table(d_unstab, d_stab, n, W, H, R, C, tv, d_crop, d_status, (hipStream_t)stream);
}

int mf_warp_u8c3(const uint8_t* d_frames, uint8_t* d_out, const void* d_table, int n, int W, int H,
                 int R, int C, const uint8_t border_bgr[3], int32_t* d_crop, void* stream)
{
    if (!d_frames || !d_out || !d_table || !border_bgr || !d_crop) { set_error("mf_warp_u8c3: null pointer"); return MF_ERR_INVALID_ARG; }
    if (d_frames == d_out) { set_error("mf_warp_u8c3: d_frames and d_out alias"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0) { set_error("mf_warp_u8c3: bad sizes"); return MF_ERR_INVALID_ARG; }
    const TableView tv = table_view(const_cast<void*>(d_table), n, W, H, R, C);
    return launch_warp(d_frames, d_out, tv, n, W, H, R, C, pack_border(border_bgr), d_crop, (hipStream_t)stream);
}

int mf_crop_scan_f64(const void* d_table, int n, int W, int H, int R, int C, int32_t* d_crop, void* stream)
{
    if (!d_table || !d_crop) { set_error("mf_crop_scan_f64: null pointer"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0) { set_error("mf_crop_scan_f64: bad sizes"); return MF_ERR_INVALID_ARG; }
    const TableView tv = table_view(const_cast<void*>(d_table), n, W, H, R, C);
    return launch_crop_scan(tv, n, W, H, R, C, d_crop, (hipStream_t)stream);
}

// ---- the same three calls with the clip-level rectangle in the CALLER's d_bounds[4] instead of inside the table blob ----

int mf_cell_table_bounds_f64(const double* d_unstab, const double* d_stab, int n, int W, int H, int R, int C,
                             void* d_table, int32_t* d_crop, int32_t* d_status, int32_t* d_bounds, void* stream)
{
    if (!d_unstab || !d_stab || !d_table || !d_crop || !d_status || !d_bounds) { set_error("mf_cell_table_bounds_f64: null pointer"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0) { set_error("mf_cell_table_bounds_f64: bad sizes"); return MF_ERR_INVALID_ARG; }
    TableView tv = table_view(d_table, n, W, H, R, C);
    tv.bounds = d_bounds;
    return launch_cell_table(d_unstab, d_stab, n, W, H, R, C, tv, d_crop, d_status, (hipStream_t)stream);
}

int mf_warp_bounds_u8c3(const uint8_t* d_frames, uint8_t* d_out, const void* d_table, int n, int W, int H,
                        int R, int C, const uint8_t border_bgr[3], int32_t* d_crop, int32_t* d_bounds, void* stream)
{
    if (!d_frames || !d_out || !d_table || !border_bgr || !d_crop || !d_bounds) { set_error("mf_warp_bounds_u8c3: null pointer"); return MF_ERR_INVALID_ARG; }
    if (d_frames == d_out) { set_error("mf_warp_bounds_u8c3: d_frames and d_out alias"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0) { set_error("mf_warp_bounds_u8c3: bad sizes"); return MF_ERR_INVALID_ARG; }
    TableView tv = table_view(const_cast<void*>(d_table), n, W, H, R, C);
    tv.bounds = d_bounds;
    return launch_warp(d_frames, d_out, tv, n, W, H, R, C, pack_border(border_bgr), d_crop, (hipStream_t)stream);
}

int mf_crop_scan_bounds_f64(const void* d_table, int n, int W, int H, int R, int C, int32_t* d_crop, int32_t* d_bounds, void* stream)
{
    if (!d_table || !d_crop || !d_bounds) { set_error("mf_crop_scan_bounds_f64: null pointer"); return MF_ERR_INVALID_ARG; }
    if (n <= 0 || R <= 0 || C <= 0) { set_error("mf_crop_scan_bounds_f64: bad sizes"); return MF_ERR_INVALID_ARG; }
    TableView tv = table_view(const_cast<void*>(d_table), n, W, H, R, C);
    tv.bounds = d_bounds;
    return launch_crop_scan(tv, n, W, H, R, C, d_crop, (hipStream_t)stream);
}

int mf_crop_reduce(const int32_t* d_crop, int n, int W, int H, int32_t* d_bounds, void* stream)
{
    if (!d_crop || !d_bounds) { set_error("mf_crop_reduce: null pointer"); return MF_ERR_INVALID_ARG; }
    return launch_crop_reduce(d_crop, n, W, H, d_bounds, (hipStream_t)stream);
}

size_t mf_crop_resize_workspace_bytes(int W, int H)
{
    return (W > 0 && H > 0) ? crop_resize_workspace_bytes(W, H) : 0;
}

int mf_crop_resize_u8c3(const uint8_t* d_frames, uint8_t* d_out, int n, int W, int H, int left, int top, int right,
                        int bottom, void* d_work, void* stream)
{
    if (!d_frames || !d_out || !d_work) { set_error("mf_crop_resize_u8c3: null pointer"); return MF_ERR_INVALID_ARG; }
    if (d_frames == d_out) { set_error("mf_crop_resize_u8c3: d_frames and d_out alias"); return MF_ERR_INVALID_ARG; }
    return launch_crop_resize(d_frames, d_out, n, W, H, left, top, right, bottom, d_work, (hipStream_t)stream);
}

size_t mf_vertex_motion_workspace_bytes(int total_features, int max_per_pair, int P, int R, int C)
{
    if (total_features < 0 || max_per_pair < 0 || P < 0 || R <= 0 || C <= 0) return 0;
    return vertex_motion_workspace_bytes(total_features, max_per_pair, P, R, C);
}

int mf_vertex_motion_f64(const double* d_early, const double* d_late, const int32_t* d_offsets, const double* d_hom,
                         int P, int total_features, int max_per_pair, int W, int H, int R, int C,
                         int ellipse_rows, int ellipse_cols, float* d_velocities, double* d_displacements,
                         void* d_work, int32_t* d_status, void* stream)
{
    if (!d_offsets || !d_displacements || !d_work || !d_status || (P > 0 && (!d_hom || !d_velocities)) ||
        (total_features > 0 && (!d_early || !d_late))) {
        set_error("mf_vertex_motion_f64: null pointer");
        return MF_ERR_INVALID_ARG;
    }
    return launch_vertex_motion(d_early, d_late, d_offsets, d_hom, P, total_features, max_per_pair, W, H, R, C,
                                ellipse_rows, ellipse_cols, d_velocities, d_displacements, d_work, d_status, (hipStream_t)stream);
}

int mf_stability_score_f64(const double* d_stab, int F, int S, double* d_series, double* d_score, void* stream)
{
    if (!d_stab || !d_series || !d_score) { set_error("mf_stability_score_f64: null pointer"); return MF_ERR_INVALID_ARG; }
    return launch_stability_score(d_stab, F, S, d_series, d_score, (hipStream_t)stream);
}

static int run_selftest(int (*launch)(unsigned long long, unsigned long long, unsigned long long*, hipStream_t),
                        uint64_t n, uint64_t seed, uint64_t* mismatches)
{
    void* d = nullptr;
    MF_HIP_TRY(hipMalloc(&d, sizeof(uint64_t)));
    hipError_t e = hipMemset(d, 0, sizeof(uint64_t));
    int rc = e == hipSuccess ? launch(n, seed, (unsigned long long*)d, nullptr) : hip_fail(e, "hipMemset");
    if (rc == MF_OK) rc = hip_fail(hipMemcpy(mismatches, d, sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(d);
    return rc;
}

int mf_selftest_sqrt(uint64_t n, uint64_t seed, uint64_t* mismatches)
{
    if (!mismatches) { set_error("mf_selftest_sqrt: null"); return MF_ERR_INVALID_ARG; }
    return run_selftest(launch_selftest_sqrt, n, seed, mismatches);
}

int mf_selftest_recip(uint64_t n, uint64_t seed, uint64_t* mismatches)
{
    if (!mismatches) { set_error("mf_selftest_recip: null"); return MF_ERR_INVALID_ARG; }
    return run_selftest(launch_selftest_recip, n, seed, mismatches);
}

int mf_selftest_fast64(uint64_t n, uint64_t seed, uint64_t* counters)
{
    if (!counters) { set_error("mf_selftest_fast64: null"); return MF_ERR_INVALID_ARG; }
    void* d = nullptr;
    MF_HIP_TRY(hipMalloc(&d, 3 * sizeof(uint64_t)));
    hipError_t e = hipMemset(d, 0, 3 * sizeof(uint64_t));
    int rc = e == hipSuccess ? launch_selftest_fast64(n, seed, (unsigned long long*)d, nullptr) : hip_fail(e, "hipMemset");
    if (rc == MF_OK) rc = hip_fail(hipMemcpy(counters, d, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(d);
    return rc;
}

int mf_selftest_fast64_margin(uint64_t n, uint64_t seed, double* max_ulps)
{
    if (!max_ulps) { set_error("mf_selftest_fast64_margin: null"); return MF_ERR_INVALID_ARG; }
    void* d = nullptr;
    MF_HIP_TRY(hipMalloc(&d, 4 * sizeof(uint64_t)));
    hipError_t e = hipMemset(d, 0, 4 * sizeof(uint64_t));
    int rc = e == hipSuccess ? launch_selftest_fast64(n, seed, (unsigned long long*)d, nullptr, (unsigned long long*)d + 3) : hip_fail(e, "hipMemset");
    if (rc == MF_OK) rc = hip_fail(hipMemcpy(max_ulps, (const uint64_t*)d + 3, sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(d);
    return rc;
}

// ---- host-buffer wrappers ------------------------------------------------------------------------

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
};
struct Stream {
    hipStream_t s = nullptr;
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
};
struct Event {
    hipEvent_t e = nullptr;
    ~Event() { if (e) (void)hipEventDestroy(e); }
};
}  // namespace

int mf_jacobi_f64_host(const double* b, double* x, const double* taps, const double* lam,
                       const double* inv_on, int F, int S, int omega, int iters, float* kernel_ms)
{
    if (!b || !x || !taps || !lam || !inv_on) { set_error("mf_jacobi_f64_host: null pointer"); return MF_ERR_INVALID_ARG; }
    if (F <= 0 || S <= 0 || omega <= 0) { set_error("mf_jacobi_f64_host: bad sizes"); return MF_ERR_INVALID_ARG; }
    const size_t nb = (size_t)F * S * sizeof(double);
    DevBuf db, dx, dt, dl, di;
    Stream st; Event e0, e1;
    MF_HIP_TRY(hipStreamCreate(&st.s));
    MF_HIP_TRY(hipEventCreate(&e0.e));
    MF_HIP_TRY(hipEventCreate(&e1.e));
    MF_HIP_TRY(db.alloc(nb)); MF_HIP_TRY(dx.alloc(nb));
    MF_HIP_TRY(dt.alloc((2 * omega + 1) * sizeof(double)));
    MF_HIP_TRY(dl.alloc(F * sizeof(double))); MF_HIP_TRY(di.alloc(F * sizeof(double)));
    MF_HIP_TRY(hipMemcpyAsync(db.p, b, nb, hipMemcpyHostToDevice, st.s));
    MF_HIP_TRY(hipMemcpyAsync(dt.p, taps, (2 * omega + 1) * sizeof(double), hipMemcpyHostToDevice, st.s));
    MF_HIP_TRY(hipMemcpyAsync(dl.p, lam, F * sizeof(double), hipMemcpyHostToDevice, st.s));
    MF_HIP_TRY(hipMemcpyAsync(di.p, inv_on, F * sizeof(double), hipMemcpyHostToDevice, st.s));
    MF_HIP_TRY(hipEventRecord(e0.e, st.s));
    int rc = mf_jacobi_f64((const double*)db.p, (double*)dx.p, (const double*)dt.p, (const double*)dl.p,
                           (const double*)di.p, F, S, omega, iters, st.s);
    if (rc != MF_OK) return rc;
    MF_HIP_TRY(hipEventRecord(e1.e, st.s));
    MF_HIP_TRY(hipMemcpyAsync(x, dx.p, nb, hipMemcpyDeviceToHost, st.s));
    MF_HIP_TRY(hipStreamSynchronize(st.s));
    if (kernel_ms) MF_HIP_TRY(hipEventElapsedTime(kernel_ms, e0.e, e1.e));
    return MF_OK;
}

}  // extern "C"

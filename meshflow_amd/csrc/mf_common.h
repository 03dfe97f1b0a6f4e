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

// Launchers (defined next to their kernels).
int launch_jacobi(const double* b, double* x, const double* taps, const double* lam, const double* inv_on,
                  int F, int S, int omega, int iters, hipStream_t st);
int launch_cell_table(const double* unstab, const double* stab, int n, int W, int H, int R, int C,
                      double* records, CellBox* boxes, int32_t* crop, int32_t* status, hipStream_t st);
int launch_warp(const uint8_t* frames, uint8_t* out, const double* records, const CellBox* boxes, int n,
                int W, int H, int R, int C, uint32_t border, int32_t* crop, hipStream_t st);
int launch_crop_reduce(const int32_t* crop, int n, int W, int H, int32_t* bounds, hipStream_t st);
int launch_crop_resize(const uint8_t* frames, uint8_t* out, int n, int W, int H, int left, int top, int right,
                       int bottom, hipStream_t st);

}  // namespace mf

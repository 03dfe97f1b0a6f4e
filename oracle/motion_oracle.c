/* CPU oracle, C restatement of the vertex-motion accumulation.  TEST INFRASTRUCTURE ONLY.
 *
 * Same arithmetic as oracle/motion_oracle.py (which is pinned against the reference, see its header), written
 * for clips of hundreds of frame pairs: /root/reference/meshflowstabilizer.py (mfs.py)
 *   :420        feature residual velocity = late - perspectiveTransform(early, H)      float64
 *   :426-448    ellipse around each feature -> covered vertices                        float64
 *   :338-353    per-vertex median of the covering features' residuals (statistics.median), 0 when none
 *   :325,354-355  + global vertex velocity (float32 perspectiveTransform of the grid - grid), -> float32
 *   :359-360    3x3 median blur, replicated borders, x and y separately
 *   :281        displacement[t+1] = displacement[t] + velocity[t]                     float64
 * Built with -ffp-contract=off; one rounding per operation in the order of the Python restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int cmp_double(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

static double median_or_zero(double* v, int n)
{
    if (n == 0) return 0.0;
    qsort(v, (size_t)n, sizeof(double), cmp_double);
    return (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0;
}

static float median9(float* v)
{
    for (int i = 1; i < 9; ++i) {                      /* insertion sort */
        const float t = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
        v[j + 1] = t;
    }
    return v[4];
}

/* One frame pair.  early/late: K x 2 float64; m: 3x3; vel: (R+1)*(C+1)*2 float32 (smoothed).
 * Returns 0, or 1 where the reference's math.sqrt would raise (negative argument, mfs.py:444). */
static int pair_velocities(const double* early, const double* late, int K, const double* m, int W, int H, int R, int C,
                           int ell_rows, int ell_cols, float* vel)
{
    const int R1 = R + 1, C1 = C + 1, V = R1 * C1;
    int status = 0;
    double* res = (double*)malloc(sizeof(double) * 2 * (size_t)(K > 0 ? K : 1));
    int* span = (int*)malloc(sizeof(int) * 2 * (size_t)R1 * (size_t)(K > 0 ? K : 1));    /* [k][row] -> first, last col */
    int* count = (int*)calloc((size_t)V + 1, sizeof(int));
    const double half_rows = (double)ell_rows / 2.0;
    for (int k = 0; k < K; ++k) {
        const double x = early[2 * k], y = early[2 * k + 1];
        const double w = (x * m[6] + y * m[7]) + m[8];
        double tx = 0.0, ty = 0.0;
        if (fabs(w) > (double)FLT_EPSILON) {
            const double iw = 1.0 / w;
            tx = ((x * m[0] + y * m[1]) + m[2]) * iw;
            ty = ((x * m[3] + y * m[4]) + m[5]) * iw;
        }
        res[2 * k] = late[2 * k] - tx;
        res[2 * k + 1] = late[2 * k + 1] - ty;
        const double frow = (y / (double)H) * (double)R;
        const double fcol = (x / (double)W) * (double)C;
        const double lo = frow - half_rows, hi = frow + half_rows;
        for (int r = 0; r < R1; ++r) {
            int first = 1, last = 0;
            if (lo <= (double)r && (double)r <= hi) {            /* ceil(lo) <= r <= floor(hi) */
                const double q = ((double)r - frow) / (double)ell_rows;
                const double s = 0.25 - q * q;
                if (s < 0.0) status = 1;
                else {
                    const double hw = (double)ell_cols * sqrt(s);
                    const double a = fcol - hw, b = fcol + hw;
                    first = a <= 0.0 ? 0 : (a > (double)C ? C + 1 : (int)ceil(a));
                    last = b >= (double)C ? C : (b < 0.0 ? -1 : (int)floor(b));
                }
            }
            span[2 * ((size_t)k * R1 + r)] = first;
            span[2 * ((size_t)k * R1 + r) + 1] = last;
            for (int c = first; c <= last; ++c) ++count[r * C1 + c];
        }
    }
    /* bucket the residuals per vertex */
    int* start = (int*)malloc(sizeof(int) * ((size_t)V + 1));
    start[0] = 0;
    for (int v = 0; v < V; ++v) start[v + 1] = start[v] + count[v];
    const int total = start[V];
    double* bx = (double*)malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
    double* by = (double*)malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
    int* fill = (int*)calloc((size_t)V, sizeof(int));
    for (int k = 0; k < K; ++k)
        for (int r = 0; r < R1; ++r) {
            const int first = span[2 * ((size_t)k * R1 + r)], last = span[2 * ((size_t)k * R1 + r) + 1];
            for (int c = first; c <= last; ++c) {
                const int v = r * C1 + c, at = start[v] + fill[v]++;
                bx[at] = res[2 * k];
                by[at] = res[2 * k + 1];
            }
        }
    /* medians + global motion -> float32 */
    float* raw = (float*)malloc(sizeof(float) * 2 * (size_t)V);
    for (int r = 0; r < R1; ++r)
        for (int c = 0; c < C1; ++c) {
            const int v = r * C1 + c;
            const double mx = median_or_zero(bx + start[v], count[v]);
            const double my = median_or_zero(by + start[v], count[v]);
            const float gxf = (float)ceil((double)(W - 1) * ((double)c / (double)C));
            const float gyf = (float)ceil((double)(H - 1) * ((double)r / (double)R));
            const double gx = (double)gxf, gy = (double)gyf;
            const double w = (gx * m[6] + gy * m[7]) + m[8];
            float px = 0.0f, py = 0.0f;
            if (fabs(w) > (double)FLT_EPSILON) {
                const double iw = 1.0 / w;
                px = (float)(((gx * m[0] + gy * m[1]) + m[2]) * iw);
                py = (float)(((gx * m[3] + gy * m[4]) + m[5]) * iw);
            }
            const float globx = px - gxf, globy = py - gyf;
            raw[2 * v] = (float)((double)globx + mx);
            raw[2 * v + 1] = (float)((double)globy + my);
        }
    /* 3x3 median blur, replicated borders */
    for (int r = 0; r < R1; ++r)
        for (int c = 0; c < C1; ++c)
            for (int comp = 0; comp < 2; ++comp) {
                float win[9];
                int n = 0;
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int rr = r + dr < 0 ? 0 : (r + dr > R ? R : r + dr);
                        const int cc = c + dc < 0 ? 0 : (c + dc > C ? C : c + dc);
                        win[n++] = raw[2 * (rr * C1 + cc) + comp];
                    }
                vel[2 * (r * C1 + c) + comp] = median9(win);
            }
    free(raw); free(fill); free(by); free(bx); free(start); free(count); free(span); free(res);
    return status;
}

/* Whole clip: P frame pairs, features concatenated (offsets[P+1]); vel [P][V][2] float32, disp [P+1][V][2] float64. */
int mfo_vertex_motion(const double* early, const double* late, const int32_t* offsets, const double* hom, int P,
                      int W, int H, int R, int C, int ell_rows, int ell_cols, float* vel, double* disp)
{
    const int V2 = (R + 1) * (C + 1) * 2;
    int status = 0;
#pragma omp parallel for schedule(dynamic) reduction(| : status)
    for (int p = 0; p < P; ++p)
        status |= pair_velocities(early + 2 * (size_t)offsets[p], late + 2 * (size_t)offsets[p], offsets[p + 1] - offsets[p],
                                  hom + 9 * (size_t)p, W, H, R, C, ell_rows, ell_cols, vel + (size_t)p * V2);
    memset(disp, 0, sizeof(double) * (size_t)V2);
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < V2; ++i)
            disp[(size_t)(p + 1) * V2 + i] = disp[(size_t)p * V2 + i] + (double)vel[(size_t)p * V2 + i];
    return status;
}

// frame.hip -- Frame post-processing of the front-end's outputs for gfx950:
//   k_undistort          Frame::UndistortKeyPoints (reference src/Frame.cc:1701-1731): cv::undistortPoints,
//                        OpenCV 3.2.0 fixed-point iteration in fp64, one thread per key point
//   k_point_cells        Frame::PosInGrid (1680-1690): grid cell of every undistorted key point
//   k_line_cells         ORB_SLAM2::LineIterator (src/lineIterator.cpp:34-76) over the 64x48 grid, one thread
//                        per key line, (cell, line) pairs in visiting order
//   k_cells_to_csr       Frame::AssignFeaturesToGrid / AssignFeaturesToGridForLine (832-872) as CSR: one
//                        workgroup, a thread owns a few cells and walks the item list in push order, so the
//                        per-cell order is the reference's without sorting or atomics
// Same operation order as oracle/frame.c (-ffp-contract=off).
#include "hvo_internal.hpp"
#include <string.h>

#define GRID_COLS 64
#define GRID_ROWS 48
#define GRID_CELLS (GRID_COLS * GRID_ROWS)
#define LINE_CELL_CAP 128          // cells one key line can visit: <= max(64, 48) + 2

__global__ __launch_bounds__(256) void k_undistort(const hvo_keypoint *__restrict__ kp, int n_fixed, const int *__restrict__ n_ptr, double fx, double fy, double cx, double cy,
                                                   double k0, double k1, double p1, double p2, double k4, hvo_keypoint *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_ptr ? *n_ptr : n_fixed;                   // the count may only exist on the device (streamed mode)
    if (i >= n) return;
    hvo_keypoint k = kp[i];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = k.x, y = k.y;
    x = (x - cx) * ifx; y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    k.x = (float)(xx * ww); k.y = (float)(yy * ww);
    out[i] = k;
}

__global__ __launch_bounds__(256) void k_point_cells(const hvo_keypoint *__restrict__ kp, const int *__restrict__ n_ptr, int n_fixed, float minx, float miny, float winv, float hinv,
                                                     int *__restrict__ cell)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    if (i >= n_fixed) return;
    if (i >= n) { cell[i] = -1; return; }
    const int px = (int)roundf(__fmul_rn(__fsub_rn(kp[i].x, minx), winv)), py = (int)roundf(__fmul_rn(__fsub_rn(kp[i].y, miny), hinv));
    cell[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
}

__global__ __launch_bounds__(64) void k_line_cells(const hvo_keyline *__restrict__ kl, const int *__restrict__ n_ptr, int n_fixed, float winv, float hinv, int *__restrict__ cell)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    if (i >= n_fixed) return;
    if (i >= n) { for (int m = 0; m < LINE_CELL_CAP; m++) cell[(size_t)i * LINE_CELL_CAP + m] = -1; return; }
    double x1 = (double)__fmul_rn(kl[i].sx, winv), y1 = (double)__fmul_rn(kl[i].sy, hinv);
    double x2 = (double)__fmul_rn(kl[i].ex, winv), y2 = (double)__fmul_rn(kl[i].ey, hinv);
    const bool steep = fabs(y2 - y1) > fabs(x2 - x1);
    double t;
    if (steep) { t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
    if (x1 > x2) { t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    const double dx = x2 - x1, dy = fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int x = (int)x1, y = (int)y1, m = 0;
    const int maxX = (int)x2;
    int *out = cell + (size_t)i * LINE_CELL_CAP;
    while (x <= maxX && m < LINE_CELL_CAP) {
        const int gx = steep ? y : x, gy = steep ? x : y;
        out[m++] = (gx < 0 || gx >= GRID_COLS || gy < 0 || gy >= GRID_ROWS) ? -1 : gx * GRID_ROWS + gy;
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
        x++;
    }
    for (; m < LINE_CELL_CAP; m++) out[m] = -1;
}

// items: n_items cell ids in push order (-1 = not assigned); item i reports index i / per_item.
// One workgroup of 1024 threads: count per cell (LDS atomics), block scan (a thread owns 3 cells), fill in arrival order, then every
// cell's short list is sorted by item number -- which IS the reference's push order -- by the thread that owns the cell.  (The first
// formulation had every thread walk the whole item list twice: 4 ms for the 25 600 slots of 200 key lines.)
__global__ __launch_bounds__(1024) void k_cells_to_csr(const int *__restrict__ cell, int n_items, int per_item,
                                                       int *__restrict__ cell_start, int *__restrict__ cell_items, int cap, int *__restrict__ total_out)
{
    __shared__ int cnt[GRID_CELLS], cur[GRID_CELLS];
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c0 = tid * 3;
    for (int q = 0; q < 3; q++) cnt[c0 + q] = 0;
    __syncthreads();
    for (int i = tid; i < n_items; i += 1024) { const int c = cell[i]; if (c >= 0) atomicAdd(&cnt[c], 1); }
    __syncthreads();
    const int n0 = cnt[c0], n1 = cnt[c0 + 1], n2 = cnt[c0 + 2];
    const int mine = n0 + n1 + n2;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; w++) { const int v = wsum[w]; if (w < wv) base += v; total += v; }
    int pos[3];
    pos[0] = base + incl - mine; pos[1] = pos[0] + n0; pos[2] = pos[1] + n1;
    for (int q = 0; q < 3; q++) { cell_start[c0 + q] = pos[q]; cur[c0 + q] = pos[q]; }
    if (tid == 1023) { cell_start[GRID_CELLS] = total; *total_out = total; }
    __syncthreads();
    for (int i = tid; i < n_items; i += 1024) {
        const int c = cell[i];
        if (c >= 0) { const int p = atomicAdd(&cur[c], 1); if (p < cap) cell_items[p] = i; }
    }
    __syncthreads();                                          // (one workgroup: its global stores are visible to it after the barrier)
    const int cnts[3] = { n0, n1, n2 };
    for (int q = 0; q < 3; q++) {
        const int s0 = pos[q], m = min(cnts[q], max(cap - s0, 0));
        for (int x = 1; x < m; x++) {                         // insertion sort by item number: a cell holds a handful of items
            const int v = cell_items[s0 + x]; int y = x - 1;
            while (y >= 0 && cell_items[s0 + y] > v) { cell_items[s0 + y + 1] = cell_items[s0 + y]; y--; }
            cell_items[s0 + y + 1] = v;
        }
        for (int x = 0; x < m; x++) cell_items[s0 + x] /= per_item;
    }
}

int frame_undistort(hvo_ctx *ctx, const hvo_keypoint *kp, int n, const float *dist5, hvo_keypoint *kp_un)
{
    if (dist5[0] == 0.0f) { memcpy(kp_un, kp, (size_t)n * sizeof(hvo_keypoint)); return HVO_OK; }     // Frame.cc:1703-1707
    const size_t bk = ((size_t)n * sizeof(hvo_keypoint) + 255) & ~(size_t)255;
    char *a = (char *)hvo_call_arena(ctx, 2 * bk);                // the context's staging arena: no allocation per call
    if (!a) return HVO_ERR_HIP;
    hvo_keypoint *d_in = (hvo_keypoint *)a, *d_out = (hvo_keypoint *)(a + bk);
    HVO_HIP(hipMemcpyAsync(d_in, kp, (size_t)n * sizeof(hvo_keypoint), hipMemcpyHostToDevice, ctx->stream));
    const int rc = frame_undistort_enqueue(ctx, ctx->stream, d_in, nullptr, n, dist5, d_out);
    if (rc) return rc;
    HVO_HIP(hipMemcpyAsync(kp_un, d_out, (size_t)n * sizeof(hvo_keypoint), hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    return HVO_OK;
}

// device-resident Frame::UndistortKeyPoints: d_kp -> d_out for the first *d_n (<= n_max) key points; k1 == 0 copies (Frame.cc:1703-1707)
int frame_undistort_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keypoint *d_kp, const int *d_n, int n_max, const float *dist5, hvo_keypoint *d_out)
{
    if (n_max < 1) return HVO_OK;
    if (dist5[0] == 0.0f) { HVO_HIP(hipMemcpyAsync(d_out, d_kp, (size_t)n_max * sizeof(hvo_keypoint), hipMemcpyDeviceToDevice, st)); return HVO_OK; }
    hipLaunchKernelGGL(k_undistort, dim3((n_max + 255) / 256), dim3(256), 0, st, d_kp, n_max, d_n, (double)ctx->p.fx, (double)ctx->p.fy, (double)ctx->p.cx, (double)ctx->p.cy,
                       (double)dist5[0], (double)dist5[1], (double)dist5[2], (double)dist5[3], (double)dist5[4], d_out);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

__global__ __launch_bounds__(256) void k_gather_angles(const hvo_keypoint *__restrict__ kp, const int *__restrict__ idx, int n, float *__restrict__ angle)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) angle[i] = kp[idx[i]].angle;
}
void frame_gather_angles_enqueue(hipStream_t st, const hvo_keypoint *d_kp, const int *d_idx, int n, float *d_angle)
{
    if (n > 0) hipLaunchKernelGGL(k_gather_angles, dim3((n + 255) / 256), dim3(256), 0, st, d_kp, d_idx, n, d_angle);
}

int frame_image_bounds(hvo_ctx *ctx, int w, int h, const float *dist5, float *bounds4)
{
    if (dist5[0] == 0.0f) { bounds4[0] = 0.f; bounds4[1] = (float)w; bounds4[2] = 0.f; bounds4[3] = (float)h; return HVO_OK; }
    hvo_keypoint c[4], o[4];
    memset(c, 0, sizeof(c));
    c[1].x = (float)w; c[2].y = (float)h; c[3].x = (float)w; c[3].y = (float)h;
    const int rc = frame_undistort(ctx, c, 4, dist5, o);
    if (rc) return rc;
    bounds4[0] = fminf(o[0].x, o[2].x); bounds4[1] = fmaxf(o[1].x, o[3].x);
    bounds4[2] = fminf(o[0].y, o[1].y); bounds4[3] = fmaxf(o[2].y, o[3].y);
    return HVO_OK;
}

// ---- the two grids, device-resident: key points / key lines and their counts (d_n, capped by n_max; or n_max when null) in HBM.
// d_cell: scratch of n_max (points) or n_max * 128 (lines) ints; d_start: 64*48+1 ints; d_items: cap ints; d_total: 1 int
// (*d_total > cap means the item list was truncated).  Nothing is allocated, nothing synchronises.
size_t frame_grid_scratch_ints(int n_max, bool lines) { return (size_t)n_max * (lines ? LINE_CELL_CAP : 1); }
int frame_points_grid_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keypoint *d_kp_un, const int *d_n, int n_max, const float *b,
                              int *d_cell, int32_t *d_start, int32_t *d_items, int cap, int *d_total)
{
    const float winv = (float)GRID_COLS / (b[1] - b[0]), hinv = (float)GRID_ROWS / (b[3] - b[2]);
    if (n_max > 0) hipLaunchKernelGGL(k_point_cells, dim3((n_max + 255) / 256), dim3(256), 0, st, d_kp_un, d_n, n_max, b[0], b[2], winv, hinv, d_cell);
    hipLaunchKernelGGL(k_cells_to_csr, dim3(1), dim3(1024), 0, st, d_cell, n_max, 1, d_start, d_items, cap, d_total);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}
int frame_lines_grid_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keyline *d_kl, const int *d_n, int n_max, const float *b,
                             int *d_cell, int32_t *d_start, int32_t *d_items, int cap, int *d_total)
{
    const float winv = (float)GRID_COLS / (b[1] - b[0]), hinv = (float)GRID_ROWS / (b[3] - b[2]);
    if (n_max > 0) hipLaunchKernelGGL(k_line_cells, dim3((n_max + 63) / 64), dim3(64), 0, st, d_kl, d_n, n_max, winv, hinv, d_cell);
    hipLaunchKernelGGL(k_cells_to_csr, dim3(1), dim3(1024), 0, st, d_cell, n_max * LINE_CELL_CAP, LINE_CELL_CAP, d_start, d_items, cap, d_total);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

// host-array forms: thin wrappers over the enqueue functions through the context's staging arena
static int grid_host(hvo_ctx *ctx, bool lines, const void *items, int n, size_t item_bytes, const float *b, int32_t *cell_start, int32_t *cell_items, int cap, int *n_out)
{
    const size_t b_in = ((size_t)n * item_bytes + 255) & ~(size_t)255, b_cell = (frame_grid_scratch_ints(n, lines) * 4 + 255) & ~(size_t)255,
                 b_start = ((GRID_CELLS + 1) * 4 + 255) & ~(size_t)255, b_items = ((size_t)(cap > 0 ? cap : 1) * 4 + 255) & ~(size_t)255;
    char *a = (char *)hvo_call_arena(ctx, b_in + b_cell + b_start + b_items + 256);
    if (!a) return HVO_ERR_HIP;
    int *d_cell = (int *)(a + b_in); int32_t *d_start = (int32_t *)(a + b_in + b_cell), *d_items = (int32_t *)(a + b_in + b_cell + b_start);
    int *d_total = (int *)(a + b_in + b_cell + b_start + b_items);
    hipStream_t st = ctx->stream;
    HVO_HIP(hipMemcpyAsync(a, items, (size_t)n * item_bytes, hipMemcpyHostToDevice, st));
    const int rc = lines ? frame_lines_grid_enqueue(ctx, st, (const hvo_keyline *)a, nullptr, n, b, d_cell, d_start, d_items, cap, d_total)
                         : frame_points_grid_enqueue(ctx, st, (const hvo_keypoint *)a, nullptr, n, b, d_cell, d_start, d_items, cap, d_total);
    if (rc) return rc;
    int total = 0;
    HVO_HIP(hipMemcpyAsync(cell_start, d_start, (GRID_CELLS + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipStreamSynchronize(st));
    if (total > 0) HVO_HIP(hipMemcpy(cell_items, d_items, (size_t)(total < cap ? total : cap) * sizeof(int), hipMemcpyDeviceToHost));
    *n_out = total;
    return total > cap ? HVO_ERR_CAPACITY : HVO_OK;
}
int frame_points_to_grid(hvo_ctx *ctx, const hvo_keypoint *kp_un, int n, const float *b, int32_t *cell_start, int32_t *cell_items, int *n_out)
{
    return grid_host(ctx, false, kp_un, n, sizeof(hvo_keypoint), b, cell_start, cell_items, n, n_out);
}
int frame_lines_to_grid(hvo_ctx *ctx, const hvo_keyline *kl, int n, const float *b, int32_t *cell_start, int32_t *cell_items, int cap, int *n_out)
{
    return grid_host(ctx, true, kl, n, sizeof(hvo_keyline), b, cell_start, cell_items, cap, n_out);
}

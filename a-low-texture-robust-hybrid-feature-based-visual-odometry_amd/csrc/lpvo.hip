// lpvo.hip -- Manhattan::computeNormalsLPVO (reference src/Manhattan.cpp:237-393; called from the RGB-D Frame constructor through
// Frame::ExtractMainImgPtNormals, src/Frame.cc:222, 1324-1329, until the coarse Manhattan frame exists): the second half of SURVEY.md 8f.4.
//
// WHICH reading.  As compiled, the reference hands this function the raw CV_16U depth image and reads it with at<float>, and its
// removeMatRow / removeMatCol take a branch that moves width * sizeof(float) bytes per row of CV_64F integral images: what it computes
// is not what it means to.  This file implements the INTENDED reading, the one
// oracle/planes_tail.c orc_normals_lpvo restates: depth in metres as CV_32F (raw * depth_map_factor), integral images with the zero row
// and column removed.  A maintainer who fixes the two lines in the reference gets these numbers; the unfixed binary's output is
// undefined behaviour (reads past every row) and is not reproduced.
//
// Kernels: k_lpvo_maps (vertex map, validity mask, central-difference tangents; a thread per pixel), k_lpvo_rows / k_lpvo_cols (the seven
// integral images in cv::integral's summation order: a running double sum along each row -- a thread per (image, row) -- then down each
// column -- a thread per (image, column); both chains are order-dependent only along their own direction), k_lpvo_sample (the 10 x 10
// box averages at stride 15, cross product, normalisation).  Same operations in the same order as the oracle (-ffp-contract=off).
#include "hvo_internal.hpp"
#include <float.h>
#include <algorithm>

#define LP_CELL 10
#define LP_STEP 15

struct LpArgs {
    const uint16_t *depth; int pitch, w, h;
    float fx, fy, cx, cy, dfac;
    float *Z, *V, *T;            // T: 7 images (u tangent xyz, v tangent xyz, mask), w*h floats each
    double *I;                   // 7 inclusive integral images
    double *normals; float *dz; int *pixel; int *count; int cap;
};

__global__ __launch_bounds__(256) void k_lpvo_maps(LpArgs a)
{
    const int w = a.w, h = a.h; const size_t N = (size_t)w * h;
    const float invfx = __fdiv_rn(1.0f, a.fx), invfy = __fdiv_rn(1.0f, a.fy);
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
        const int v = (int)(i / w), u = (int)(i - (size_t)v * w);
        auto zat = [&](int vv, int uu) { return __fmul_rn((float)a.depth[(size_t)vv * a.pitch + uu], a.dfac); };
        auto vert = [&](int vv, int uu, float p[3]) {
            const float z = zat(vv, uu);
            p[0] = p[1] = p[2] = 0.f;
            if (z > 0.2f && z < 7.0f) { p[0] = __fmul_rn(__fmul_rn(__fsub_rn((float)uu, a.cx), z), invfx); p[1] = __fmul_rn(__fmul_rn(__fsub_rn((float)vv, a.cy), z), invfy); p[2] = z; }
        };
        float t[7] = { 0, 0, 0, 0, 0, 0, 0 };
        if (u >= 1 && u < w - 1 && v >= 1 && v < h - 1) {
            const float zc = zat(v, u), zl = zat(v, u - 1), zr = zat(v, u + 1), zu = zat(v - 1, u), zd = zat(v + 1, u);
#define LP_BAD(q) ((q) < 0.2f || (q) > 7.0f)
            if (!(LP_BAD(zc) || LP_BAD(zl) || LP_BAD(zr) || LP_BAD(zu) || LP_BAD(zd))) {
#undef LP_BAD
                float pl[3], pr[3], pu[3], pd[3];
                vert(v, u - 1, pl); vert(v, u + 1, pr); vert(v - 1, u, pu); vert(v + 1, u, pd);
                t[6] = 1.0f;
                for (int k = 0; k < 3; k++) { t[k] = __fsub_rn(pr[k], pl[k]); t[3 + k] = __fsub_rn(pd[k], pu[k]); }
            }
        }
        for (int k = 0; k < 7; k++) a.T[(size_t)k * N + i] = t[k];
        float pc[3]; vert(v, u, pc);
        a.Z[i] = pc[2];                                          // vertexMap(v, u)[2]: z when 0.2 < z < 7, else 0
    }
}

// running double sums along the rows: I(y, x) = sum of T(y, 0..x), a thread per (image, row)
__global__ __launch_bounds__(256) void k_lpvo_rows(LpArgs a)
{
    const int w = a.w, h = a.h; const size_t N = (size_t)w * h;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= 7 * h) return;
    const int k = id / h, y = id - k * h;
    const float *src = a.T + (size_t)k * N + (size_t)y * w;
    double *dst = a.I + (size_t)k * N + (size_t)y * w;
    double s = 0;
    for (int x = 0; x < w; x++) { s += (double)src[x]; dst[x] = s; }
}

// ... then down the columns: I(y, x) = I(y-1, x) + rowsum(y, x), a thread per (image, column)
__global__ __launch_bounds__(256) void k_lpvo_cols(LpArgs a)
{
    const int w = a.w, h = a.h; const size_t N = (size_t)w * h;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= 7 * w) return;
    const int k = id / w, x = id - k * w;
    double *col = a.I + (size_t)k * N + x;
    double up = 0;
    for (int y = 0; y < h; y++) { const double v = up + col[(size_t)y * w]; col[(size_t)y * w] = v; up = v; }
}

__global__ __launch_bounds__(256) void k_lpvo_sample(LpArgs a, int nu, int nv)
{
    const int w = a.w; const size_t N = (size_t)w * a.h;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= nu * nv) return;
    const int v = LP_CELL + (id / nu) * LP_STEP, u = LP_CELL + (id % nu) * LP_STEP;
    const size_t i = (size_t)v * w + u;
    // the reference emits the samples in raster order of the grid: rank of this sample among the valid ones = a prefix count; the grid
    // is small (<= ~1400 positions), so every thread counts its predecessors itself
    if (a.T[6 * N + i] != 1.0f) return;
    int rank = 0;
    for (int q = 0; q < id; q++) { const int vq = LP_CELL + (q / nu) * LP_STEP, uq = LP_CELL + (q % nu) * LP_STEP; rank += a.T[6 * N + (size_t)vq * w + uq] == 1.0f; }
    atomicAdd(a.count, 1);
    if (rank >= a.cap) return;
    auto box = [&](int k) {
        const double *I = a.I + (size_t)k * N;
        return I[i] - I[i - (size_t)LP_CELL * w] - I[i - LP_CELL] + I[i - (size_t)LP_CELL * w - LP_CELL];
    };
    const int numPts = (int)box(6);
    const double uv[3] = { box(0) / numPts, box(1) / numPts, box(2) / numPts }, vv[3] = { box(3) / numPts, box(4) / numPts, box(5) / numPts };
    const double nx = vv[1] * uv[2] - vv[2] * uv[1], ny = vv[2] * uv[0] - vv[0] * uv[2], nz = vv[0] * uv[1] - vv[1] * uv[0];
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    const double sc = len > DBL_EPSILON ? 1.0 / len : 0.0;       // cv::normalize (NORM_L2, alpha = 1)
    a.normals[3 * rank] = nx * sc; a.normals[3 * rank + 1] = ny * sc; a.normals[3 * rank + 2] = nz * sc;
    a.dz[rank] = a.Z[i]; a.pixel[2 * rank] = u; a.pixel[2 * rank + 1] = v;
}

static size_t lp_al(size_t v) { return (v + 255) & ~(size_t)255; }
size_t lpvo_scratch_bytes(int w, int h) { const size_t N = (size_t)w * h; return lp_al(N * 4) + lp_al(7 * N * 4) + lp_al(7 * N * 8) + 256; }
int lpvo_capacity(int w, int h) { return ((h - 1 - LP_CELL + LP_STEP - 1) / LP_STEP) * ((w - 1 - LP_CELL + LP_STEP - 1) / LP_STEP); }

// device-resident form: depth in HBM; scratch of lpvo_scratch_bytes; d_normals cap x 3 doubles, d_dz cap floats, d_pixel cap x 2 ints, d_count 1 int
int lpvo_enqueue(hvo_ctx *ctx, hipStream_t st, const uint16_t *d_depth, int pitch, int w, int h, void *scratch, double *d_normals, float *d_dz, int *d_pixel, int *d_count, int cap)
{
    if (w < 2 * LP_CELL + 2 || h < 2 * LP_CELL + 2) return HVO_ERR_UNSUPPORTED;
    const size_t N = (size_t)w * h;
    LpArgs a;
    a.depth = d_depth; a.pitch = pitch; a.w = w; a.h = h; a.fx = ctx->p.fx; a.fy = ctx->p.fy; a.cx = ctx->p.cx; a.cy = ctx->p.cy; a.dfac = ctx->p.depth_map_factor;
    char *s = (char *)scratch;
    a.Z = (float *)s; a.V = nullptr; a.T = (float *)(s + lp_al(N * 4)); a.I = (double *)(s + lp_al(N * 4) + lp_al(7 * N * 4));
    a.normals = d_normals; a.dz = d_dz; a.pixel = d_pixel; a.count = d_count; a.cap = cap;
    const int nv = (h - 1 - LP_CELL + LP_STEP - 1) / LP_STEP, nu = (w - 1 - LP_CELL + LP_STEP - 1) / LP_STEP;
    HVO_HIP(hipMemsetAsync(d_count, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_lpvo_maps, dim3((unsigned)std::min<size_t>((N + 255) / 256, 4096)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_lpvo_rows, dim3((7 * h + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_lpvo_cols, dim3((7 * w + 255) / 256), dim3(256), 0, st, a);
    if (nu > 0 && nv > 0) hipLaunchKernelGGL(k_lpvo_sample, dim3((nu * nv + 255) / 256), dim3(256), 0, st, a, nu, nv);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

extern "C" int hvo_normals_lpvo(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride, double *normals3, float *depth_out, int32_t *pixel2, int cap, int *n)
{
    if (!ctx || !n) return HVO_ERR_INVALID_ARG;
    *n = 0;
    if (!depth || !normals3 || !depth_out || !pixel2 || cap < 0 || w <= 0 || h <= 0 || stride < 2 * w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hipStream_t st = ctx->stream;
    const int full = lpvo_capacity(w, h);
    if (full <= 0) return HVO_ERR_UNSUPPORTED;
    const size_t b_d = lp_al((size_t)w * h * 2), b_n = lp_al((size_t)full * 24), b_z = lp_al((size_t)full * 4), b_p = lp_al((size_t)full * 8);
    char *a = (char *)hvo_call_arena(ctx, b_d + b_n + b_z + b_p + 256 + lpvo_scratch_bytes(w, h));
    if (!a) return HVO_ERR_HIP;
    uint16_t *dd = (uint16_t *)a; double *dn = (double *)(a + b_d); float *dz = (float *)(a + b_d + b_n); int *dp = (int *)(a + b_d + b_n + b_z);
    int *dc = (int *)(a + b_d + b_n + b_z + b_p); void *scratch = a + b_d + b_n + b_z + b_p + 256;
    HVO_HIP(hipMemcpy2DAsync(dd, (size_t)w * 2, depth, stride, (size_t)w * 2, h, hipMemcpyHostToDevice, st));
    const int rc = lpvo_enqueue(ctx, st, dd, w, w, h, scratch, dn, dz, dp, dc, full);
    if (rc) return rc;
    int cnt = 0;
    HVO_HIP(hipMemcpyAsync(&cnt, dc, sizeof(int), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipStreamSynchronize(st));
    *n = cnt;
    const int m = std::min(cnt, cap);
    if (m > 0) {
        HVO_HIP(hipMemcpy(normals3, dn, (size_t)m * 24, hipMemcpyDeviceToHost));
        HVO_HIP(hipMemcpy(depth_out, dz, (size_t)m * 4, hipMemcpyDeviceToHost));
        HVO_HIP(hipMemcpy(pixel2, dp, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    return cnt > cap ? HVO_ERR_CAPACITY : HVO_OK;
}

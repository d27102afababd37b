// vps.hip -- vanishing-point clustering of a frame's key lines, reference src/Frame.cc:330-337:
//   Frame::getVPHypVia2Lines   src/Frame.cc:442-545   k_vp_lines (line functions, lengths, orientations), k_vp_hyp (105 x 360 hypotheses)
//   Frame::getSphereGrids      src/Frame.cc:546-650   k_vp_pairs + k_vp_grid (90 x 360 sphere grid), k_vp_smooth
//   Frame::getBestVpsHyp       src/Frame.cc:651-707   k_vp_hyp (scores), k_vp_best (first maximum)
//   Frame::line2Vps            src/Frame.cc:708-778   k_vp_best (cluster of every line)
// Determinism: the reference draws its line pairs from a time-seeded rand(); here the caller passes a seed and hypothesis
// group i draws from its own xorshift32 stream (seed, i) -- the rule of oracle/vps.c, which this file follows step by step.
// The sphere grid is accumulated in the reference's order (pairs (i, j), i-major): a cell's owner thread walks the pair list
// in that order, so the sums are the sequential ones whatever the launch shape (fp64 addition is not associative, and the
// best hypothesis is an argmax over sums of grid cells).  sin / cos / atan / acos / atan2 are the device's.
#include "hvo_internal.hpp"
#include <cstring>
#include <cmath>
#include <hip/hip_runtime.h>

#define VP_PI 3.1415926535897932384626433832795
#define VP_NUM2 360
#define VP_LA 90
#define VP_LO 360
#define VP_CELLS (VP_LA * VP_LO)
#define VP_MAX_LINES 1024
#define VP_REDRAWS 64

static __device__ __forceinline__ void vcross(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static __device__ __forceinline__ unsigned vxs32(unsigned &s) { unsigned x = s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; s = x; return x; }

// src/Frame.cc:454-473
// (every kernel takes the line count from n_ptr when it is given: in the pipelines it only exists on the device)
__global__ __launch_bounds__(256) void k_vp_lines(const hvo_keyline *__restrict__ kl, const int *__restrict__ n_ptr, int n_fixed, double *__restrict__ para, double *__restrict__ len, double *__restrict__ ori)
{
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p1[3] = { (double)kl[i].sx, (double)kl[i].sy, 1.0 }, p2[3] = { (double)kl[i].ex, (double)kl[i].ey, 1.0 };
    double c[3];
    vcross(p1, p2, c);
    para[3 * i] = c[0]; para[3 * i + 1] = c[1]; para[3 * i + 2] = c[2];
    const double dx = (double)(kl[i].ex - kl[i].sx), dy = (double)(kl[i].ey - kl[i].sy);
    len[i] = sqrt(dx * dx + dy * dy);
    double o = atan2(dy, dx);
    if (o < 0) o += VP_PI;
    ori[i] = o;
}

// one thread per pair (i < j), written at the pair's rank in the reference's loop order: cell (-1: contributes nothing), value
__global__ __launch_bounds__(256) void k_vp_pairs(const double *__restrict__ para, const double *__restrict__ len, const double *__restrict__ ori, const int *__restrict__ n_ptr, int n_fixed,
                                                  double fx, double cx, double cy, int *__restrict__ cell, double *__restrict__ val)
{
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j <= i || j >= n) return;
    const size_t p = (size_t)i * (2 * (size_t)n - i - 1) / 2 + (j - i - 1);
    const double acc = 1.0 / 180.0 * VP_PI, tol = 60.0 / 180.0 * VP_PI;
    double pt[3];
    vcross(para + 3 * i, para + 3 * j, pt);
    int c = -1; double v = 0;
    if (pt[2] != 0) {
        const double x = pt[0] / pt[2], y = pt[1] / pt[2];
        const double X = x - cx, Y = y - cy, Z = fx, N = sqrt(X * X + Y * Y + Z * Z);
        const double latitude = acos(Z / N), longitude = atan2(X, Y) + VP_PI;
        int LA = (int)(latitude / acc); if (LA >= VP_LA) LA = VP_LA - 1;
        int LO = (int)(longitude / acc); if (LO >= VP_LO) LO = VP_LO - 1;
        double dev = fabs(ori[i] - ori[j]);
        dev = fmin(VP_PI - dev, dev);
        if (!(dev > tol)) { c = LA * VP_LO + LO; v = sqrt(len[i] * len[j]) * (sin(2.0 * dev) + 0.2); }
    }
    cell[p] = c; val[p] = v;
}

// The sphere grid: every cell adds the values of ITS pairs in pair order (fp64 addition is not associative, and the best hypothesis
// is an argmax over sums of cells).  One workgroup builds a CSR of the pair list by cell in LDS -- count (atomics), scan (a thread owns 32
// cells), fill in arrival order, sort each cell's few pairs by pair index = the reference's loop order -- and adds.  (The first
// formulation had the owner of every one of the 32 400 cells scan all 19 900 pairs: 1.1 ms.)
#define VP_CPT 32                                    // cells per thread: 1024 x 32 >= 32 400
#define VP_BIG 48                                    // a cell with more pairs than this is ordered by the whole workgroup
#define VP_BIGMAX 2048                               // such cells a frame can have (523 776 pairs / 48 would be 10 912: the rest keep the owner's sort)
__global__ __launch_bounds__(1024) void k_vp_grid(const int *__restrict__ cell, const double *__restrict__ val, const int *__restrict__ n_ptr, int n_fixed, double *__restrict__ raw,
                                                  int *__restrict__ order, int *__restrict__ order2)
{
    __shared__ int nbig, bigs[VP_BIGMAX][3];          // (cell, start, end) of the cells the workgroup orders together
    extern __shared__ int vg_cnt[];                  // VP_CPT * 1024 cursors + 16 wave sums
    int *wsum = vg_cnt + VP_CPT * 1024;
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    const int npairs = n < 2 ? 0 : n * (n - 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c0 = tid * VP_CPT;
    for (int q = 0; q < VP_CPT; q++) vg_cnt[c0 + q] = 0;
    if (tid == 0) nbig = 0;
    __syncthreads();
    for (int p = tid; p < npairs; p += 1024) { const int c = cell[p]; if (c >= 0) atomicAdd(&vg_cnt[c], 1); }
    __syncthreads();
    int mine = 0;
    for (int q = 0; q < VP_CPT; q++) mine += vg_cnt[c0 + q];
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wsum[w];
    int run = base + incl - mine;
    for (int q = 0; q < VP_CPT; q++) { const int k = vg_cnt[c0 + q]; vg_cnt[c0 + q] = run; run += k; }     // cursor = start
    __syncthreads();
    for (int p = tid; p < npairs; p += 1024) { const int c = cell[p]; if (c >= 0) order[atomicAdd(&vg_cnt[c], 1)] = p; }   // cursor ends at the cell's end
    __syncthreads();
    int s0 = base + incl - mine;
    for (int q = 0; q < VP_CPT; q++) {
        const int c = c0 + q, e0 = vg_cnt[c];
        // Many line pairs can meet in ONE sphere cell (a Manhattan scene: 100 parallel lines give ~5 k pairs at their vanishing point): the
        // owner's insertion sort is quadratic in a cell's population -- millions of dependent global accesses by one lane.  Such cells
        // are listed and ordered by the whole workgroup below (rank by counting); the owner then only adds.
        bool big = false;
        if (e0 - s0 > VP_BIG) { const int slot = atomicAdd(&nbig, 1); if (slot < VP_BIGMAX) { bigs[slot][0] = c; bigs[slot][1] = s0; bigs[slot][2] = e0; big = true; } }
        if (!big) {
            double acc = 0.0;
            for (int x = s0 + 1; x < e0; x++) {                  // insertion sort by pair index
                const int v = order[x]; int y = x - 1;
                while (y >= s0 && order[y] > v) { order[y + 1] = order[y]; y--; }
                order[y + 1] = v;
            }
            for (int x = s0; x < e0; x++) acc += val[order[x]];
            if (c < VP_CELLS) raw[c] = acc;
        }
        s0 = e0;
    }
    __syncthreads();
    const int nb = min(nbig, VP_BIGMAX);
    for (int b = 0; b < nb; b++) {
        const int bs = bigs[b][1], be = bigs[b][2];
        for (int x = bs + tid; x < be; x += 1024) {              // pair indices are unique: the rank is the position
            const int v = order[x];
            int r = 0;
            for (int y = bs; y < be; y++) r += order[y] < v;
            order2[bs + r] = v;
        }
    }
    __syncthreads();
    for (int b = tid; b < nb; b += 1024) {                       // the sums in pair order, one cell per thread
        const int c = bigs[b][0], bs = bigs[b][1], be = bigs[b][2];
        double acc = 0.0;
        for (int x = bs; x < be; x++) acc += val[order2[x]];
        if (c < VP_CELLS) raw[c] = acc;
    }
}

// src/Frame.cc:629-649: new = old + (3x3 sum) / 9 in the interior, 0 on the border rows / columns
__global__ __launch_bounds__(256) void k_vp_smooth(const double *__restrict__ raw, double *__restrict__ grid)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= VP_CELLS) return;
    const int i = c / VP_LO, j = c - i * VP_LO;
    double out = 0.0;
    if (i >= 1 && i < VP_LA - 1 && j >= 1 && j < VP_LO - 1) {
        double tot = 0.0;
        for (int m = 0; m < 3; m++) for (int q = 0; q < 3; q++) tot += raw[(i - 1 + m) * VP_LO + (j - 1 + q)];
        out = raw[c] + tot / 9;
    }
    grid[c] = out;
}

static __device__ __forceinline__ double vp_cell_of(const double *grid, const double *v)
{
    const double oneDegree = 1.0 / 180.0 * VP_PI;
    if (v[2] == 0.0) return 0.0;
    const double latitude = acos(v[2]), longitude = atan2(v[0], v[1]) + VP_PI;
    int LA = (int)(latitude / oneDegree); if (LA == 90) LA = 89;
    // a longitude within 1e-6 degree of a whole degree goes to that degree's cell: the determinism rule of oracle/vps.c vp_score (every
    // hypothesis' second direction sits on a cell boundary by construction, and the reference's choice there is libm rounding noise)
    const double lo_f = longitude / oneDegree, lo_r = nearbyint(lo_f);
    int LO = fabs(lo_f - lo_r) < 1e-6 ? (int)lo_r : (int)lo_f; if (LO >= 360) LO = 359;
    if (LA < 0 || LA > 89 || LO < 0 || LO > 359) return 0.0;     // cannot happen for unit vectors with z >= 0; keeps a NaN hypothesis from reading outside
    return grid[LA * VP_LO + LO];
}

// block = hypothesis group i (one pair of lines), thread = j of its 360 hypotheses: vp1, vp2, vp3 and the score
__global__ __launch_bounds__(384) void k_vp_hyp(const double *__restrict__ para, const int *__restrict__ n_ptr, int n_fixed, double fx, double cx, double cy, unsigned seed,
                                                const double *__restrict__ grid, double *__restrict__ hyp, double *__restrict__ score)
{
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    const int i = blockIdx.x, j = threadIdx.x;
    if (n < 2) {                                     // the reference skips the path (src/Frame.cc:328): zero hypotheses of score 0
        if (j < VP_NUM2) { const size_t h = (size_t)i * VP_NUM2 + j; for (int q = 0; q < 9; q++) hyp[h * 9 + q] = 0.0; score[h] = 0.0; }
        return;
    }
    unsigned rs = seed ^ (0x9E3779B9u * (unsigned)(i + 1)); if (rs == 0) rs = 0x6D2B79F5u;
    double vp1[3];
    // The reference redraws without bound (src/Frame.cc:487-491); with every pair of lines meeting at infinity that is a hang, on a GPU
    // an unrecoverable one.  After VP_REDRAWS draws the group gives up: 360 zero hypotheses of score 0 (the rule of oracle/vps.c).
    for (int tries = 0;; tries++) {                  // uniform over the block: every thread draws the same pair
        if (tries >= VP_REDRAWS) {
            if (j < VP_NUM2) { const size_t h = (size_t)i * VP_NUM2 + j; for (int q = 0; q < 9; q++) hyp[h * 9 + q] = 0.0; score[h] = 0.0; }
            return;
        }
        const int idx1 = (int)((vxs32(rs) & 0x7fffffffu) % (unsigned)n);
        int idx2 = (int)((vxs32(rs) & 0x7fffffffu) % (unsigned)n);
        while (idx2 == idx1) idx2 = (int)((vxs32(rs) & 0x7fffffffu) % (unsigned)n);
        double v[3];
        vcross(para + 3 * idx1, para + 3 * idx2, v);
        if (v[2] == 0) continue;
        vp1[0] = v[0] / v[2] - cx; vp1[1] = v[1] / v[2] - cy; vp1[2] = fx;
        break;
    }
    if (j >= VP_NUM2) return;
    if (vp1[2] == 0) vp1[2] = 0.0011;
    double N = sqrt(vp1[0] * vp1[0] + vp1[1] * vp1[1] + vp1[2] * vp1[2]);
    { const double s = 1.0 / N; vp1[0] *= s; vp1[1] *= s; vp1[2] *= s; }
    const double stepVp2 = 2.0 * VP_PI / VP_NUM2;
    const double lambda = j * stepVp2;
    const double k1 = vp1[0] * sin(lambda) + vp1[1] * cos(lambda), k2 = vp1[2];
    const double phi = atan(-k2 / k1);
    double vp2[3] = { sin(phi) * sin(lambda), sin(phi) * cos(lambda), cos(phi) }, vp3[3];
    if (vp2[2] == 0.0) vp2[2] = 0.0011;
    N = sqrt(vp2[0] * vp2[0] + vp2[1] * vp2[1] + vp2[2] * vp2[2]);
    { const double s = 1.0 / N; vp2[0] *= s; vp2[1] *= s; vp2[2] *= s; }
    if (vp2[2] < 0) { vp2[0] *= -1.0; vp2[1] *= -1.0; vp2[2] *= -1.0; }
    vcross(vp1, vp2, vp3);
    if (vp3[2] == 0.0) vp3[2] = 0.0011;
    N = sqrt(vp3[0] * vp3[0] + vp3[1] * vp3[1] + vp3[2] * vp3[2]);
    { const double s = 1.0 / N; vp3[0] *= s; vp3[1] *= s; vp3[2] *= s; }
    if (vp3[2] < 0) { vp3[0] *= -1.0; vp3[1] *= -1.0; vp3[2] *= -1.0; }
    const size_t h = (size_t)i * VP_NUM2 + j;
    double *o = hyp + h * 9;
    o[0] = vp1[0]; o[1] = vp1[1]; o[2] = vp1[2]; o[3] = vp2[0]; o[4] = vp2[1]; o[5] = vp2[2]; o[6] = vp3[0]; o[7] = vp3[1]; o[8] = vp3[2];
    double s = 0.0;
    s += vp_cell_of(grid, vp1); s += vp_cell_of(grid, vp2); s += vp_cell_of(grid, vp3);
    score[h] = s;
}

// first maximum above 0 (index 0 when there is none), then line2Vps: one workgroup
__global__ __launch_bounds__(1024) void k_vp_best(const double *__restrict__ score, int nh, const double *__restrict__ hyp,
                                                  const hvo_keyline *__restrict__ kl, const int *__restrict__ n_ptr, int n_fixed, double fx, double fy, double cx, double cy, double th_angle,
                                                  hvo_vp_result *__restrict__ res, int32_t *__restrict__ vp_idx)
{
    __shared__ double bs[1024]; __shared__ int bi[1024];
    const int n = n_ptr ? min(*n_ptr, n_fixed) : n_fixed;
    const int tid = threadIdx.x;
    double s = 0.0; int b = 0x7FFFFFFF;
    for (int h = tid; h < nh; h += 1024) { const double v = score[h]; if (v > s) { s = v; b = h; } }     // ascending h: the first of this thread's maxima
    bs[tid] = s; bi[tid] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            const double s2 = bs[tid + o]; const int b2 = bi[tid + o];
            if (s2 > bs[tid] || (s2 == bs[tid] && b2 < bi[tid])) { bs[tid] = s2; bi[tid] = b2; }
        }
        __syncthreads();
    }
    const int best = bi[0] == 0x7FFFFFFF ? 0 : bi[0];
    const double *v = hyp + (size_t)best * 9;
    if (tid == 0) {
        for (int q = 0; q < 9; q++) res->vps[q / 3][q % 3] = v[q];
        res->score = bs[0]; res->best = best; res->n_hypotheses = nh;
        if (n < 2) { for (int q = 0; q < 9; q++) res->vps[q / 3][q % 3] = 0.0; res->score = 0.0; res->best = 0; res->n_hypotheses = 0; }
    }
    if (n < 2) { for (int i = tid; i < n; i += 1024) vp_idx[i] = 3; return; }
    double vx[3], vy[3];
    for (int j = 0; j < 3; j++) { vx[j] = v[3 * j] * fx / v[3 * j + 2] + cx; vy[j] = v[3 * j + 1] * fy / v[3 * j + 2] + cy; }
    for (int i = tid; i < n; i += 1024) {
        const double x1 = kl[i].sx, y1 = kl[i].sy, x2 = kl[i].ex, y2 = kl[i].ey;
        const double xm = (x1 + x2) / 2.0, ym = (y1 + y2) / 2.0;
        double v1x = x1 - x2, v1y = y1 - y2;
        const double N1 = sqrt(v1x * v1x + v1y * v1y);
        v1x /= N1; v1y /= N1;
        double minAngle = 1000.0; int bj = 0;
        for (int j = 0; j < 3; j++) {
            double v2x = vx[j] - xm, v2y = vy[j] - ym;
            const double N2 = sqrt(v2x * v2x + v2y * v2y);
            v2x /= N2; v2y /= N2;
            double c = v1x * v2x + v1y * v2y;
            if (c > 1.0) c = 1.0;
            if (c < -1.0) c = -1.0;
            double angle = acos(c);
            angle = fmin(VP_PI - angle, angle);
            if (angle < minAngle) { minAngle = angle; bj = j; }
        }
        vp_idx[i] = minAngle < th_angle ? bj : 3;
    }
}

static int vp_iterations()
{
    const double noiseRatio = 0.5, p = 1.0 / 3.0 * pow(1.0 - noiseRatio, 2), confEfficience = 0.9999;
    return (int)(log(1 - confEfficience) / log(1.0 - p));
}

size_t vp_scratch_bytes(int nmax)
{
    const size_t npairs = (size_t)nmax * (nmax > 0 ? nmax - 1 : 0) / 2, nh = (size_t)vp_iterations() * VP_NUM2;
    return (5 * (size_t)nmax + npairs + 2 * VP_CELLS + 10 * nh) * sizeof(double) + 3 * npairs * sizeof(int) + 256;
}

// device-resident form: key lines and their count (d_n, capped by nmax; or nmax itself when d_n is null) already in HBM; scratch of
// vp_scratch_bytes(nmax); d_res / d_idx (nmax entries) / d_grid (optional, 90 x 360 doubles) receive the results.  Nothing is
// allocated, nothing synchronises.
int vp_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keyline *d_kl, const int *d_n, int nmax, unsigned seed, double th_angle,
               void *scratch, hvo_vp_result *d_res, int32_t *d_idx, double *d_grid_out)
{
    if (nmax < 1) return HVO_OK;
    const hvo_params &P = ctx->p;
    const int it = vp_iterations(), nh = it * VP_NUM2;
    const size_t npairs = (size_t)nmax * (nmax - 1) / 2;
    double *para = (double *)scratch, *len = para + 3 * (size_t)nmax, *ori = len + nmax, *val = ori + nmax, *raw = val + npairs, *grid = raw + VP_CELLS,
           *hyp = grid + VP_CELLS, *score = hyp + 9 * (size_t)nh;
    int *dcell = (int *)(score + nh), *dorder = dcell + npairs, *dorder2 = dorder + npairs;
    const double fx = P.fx, fy = P.fy, cx = P.cx, cy = P.cy;
    hipLaunchKernelGGL(k_vp_lines, dim3((nmax + 255) / 256), dim3(256), 0, st, d_kl, d_n, nmax, para, len, ori);
    if (nmax > 1) hipLaunchKernelGGL(k_vp_pairs, dim3((nmax + 255) / 256, nmax - 1), dim3(256), 0, st, para, len, ori, d_n, nmax, fx, cx, cy, dcell, val);
    {
        const size_t lds = (VP_CPT * 1024 + 16) * sizeof(int);            // 131 KB: one workgroup owns the CU
        if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_vp_grid), lds)) return HVO_ERR_HIP;      // per device, under the library's lock
        hipLaunchKernelGGL(k_vp_grid, dim3(1), dim3(1024), lds, st, dcell, val, d_n, nmax, raw, dorder, dorder2);
    }
    hipLaunchKernelGGL(k_vp_smooth, dim3((VP_CELLS + 255) / 256), dim3(256), 0, st, raw, grid);
    hipLaunchKernelGGL(k_vp_hyp, dim3(it), dim3(384), 0, st, para, d_n, nmax, fx, cx, cy, seed, grid, hyp, score);
    hipLaunchKernelGGL(k_vp_best, dim3(1), dim3(1024), 0, st, score, nh, hyp, d_kl, d_n, nmax, fx, fy, cx, cy, th_angle, d_res, d_idx);
    if (d_grid_out) HVO_HIP(hipMemcpyAsync(d_grid_out, grid, VP_CELLS * sizeof(double), hipMemcpyDeviceToDevice, st));
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

// host-array form: a thin wrapper over vp_enqueue through the context's staging arena (tail.hip): no allocation per call
extern "C" int hvo_vanishing_points(hvo_ctx *ctx, const hvo_keyline *kl, int n, uint32_t seed, double th_angle,
                                    hvo_vp_result *res, int32_t *vp_idx, double *grid_out)
{
    if (!ctx || !res || n < 0 || n > VP_MAX_LINES) return HVO_ERR_INVALID_ARG;
    if (n < 2) {                                       // the reference skips the path (src/Frame.cc:328): nothing is a structure line
        std::memset(res, 0, sizeof(*res));
        if (vp_idx) for (int i = 0; i < n; i++) vp_idx[i] = 3;
        return HVO_OK;
    }
    if (!kl || !vp_idx) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hipStream_t st = ctx->stream;
    const size_t b_kl = ((size_t)n * sizeof(hvo_keyline) + 255) & ~(size_t)255, b_idx = ((size_t)n * 4 + 255) & ~(size_t)255, b_grid = VP_CELLS * sizeof(double);
    char *a = (char *)hvo_call_arena(ctx, b_kl + 256 + b_idx + b_grid + vp_scratch_bytes(n));
    if (!a) return HVO_ERR_HIP;
    hvo_keyline *dk = (hvo_keyline *)a; hvo_vp_result *dres = (hvo_vp_result *)(a + b_kl); int32_t *didx = (int32_t *)(a + b_kl + 256);
    double *dgrid = (double *)(a + b_kl + 256 + b_idx); void *scratch = a + b_kl + 256 + b_idx + b_grid;
    HVO_HIP(hipMemcpyAsync(dk, kl, (size_t)n * sizeof(hvo_keyline), hipMemcpyHostToDevice, st));
    int rc = vp_enqueue(ctx, st, dk, nullptr, n, seed, th_angle, scratch, dres, didx, grid_out ? dgrid : nullptr);
    if (rc) return rc;
    HVO_HIP(hipMemcpyAsync(res, dres, sizeof(hvo_vp_result), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipMemcpyAsync(vp_idx, didx, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (grid_out) HVO_HIP(hipMemcpyAsync(grid_out, dgrid, VP_CELLS * sizeof(double), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipStreamSynchronize(st));
    return HVO_OK;
}

// planes_tail.hip -- the tail of Frame::ComputePlanes after the plane detector (reference src/Frame.cc:2110-2212) and
// Frame::MaxPointDistanceFromPlane (2214-2274) for gfx950:
//   hvo_plane_clouds      per extracted plane: its pixels' points as float, pcl::VoxelGrid(0.1 m), the distance gate, the
//                         pcl::SACSegmentation plane refit and the sign rule -> mvPlanePoints / mvPlaneCoefficients
//   hvo_surface_normals   the 1/3-resolution cloud + pcl::IntegralImageNormalEstimation(AVERAGE_3D_GRADIENT, 0.05, 10) sampled at
//                         the odd grid positions -> vSurfaceNormal
// PCL is not vendored by the reference; the semantics restated here are those of oracle/planes_tail.c (ASSUMED, PCL 1.8), which
// this file follows operation by operation (-ffp-contract=off).
//
// Kernels
//   k_pc_bbox      per pixel: the point of a labelled pixel, per-plane bounding box (ordered-int float atomics, LDS first) and pixel count
//   k_pc_setup     per plane: voxel-grid geometry (min_b, div_b), its slice of the voxel table, (n, d) as float
//   k_pc_accum     per pixel: voxel index, coordinates accumulated as 2^-24 m fixed point with 64-bit integer atomics: the centroid is
//                  the exact mean whatever the order (PCL's own float sum depends on an unstable sort, see the oracle)
//   k_pc_count / k_pc_emit   per plane: non-empty voxels in ascending index order (block scan), centroids, the distance gate
//   k_pc_refit     per plane, one workgroup: RANSAC exactly as pcl::RandomSampleConsensus runs it (boost::mt19937 seeded 12345 on one
//                  lane, partial Fisher-Yates over a persistent index array, 3-point models, inlier counts by all threads, the
//                  adaptive iteration bound), then the float covariance of the inliers in index order (one lane: float sums are
//                  order dependent) and pcl::eigen33
//   k_sn_cloud / k_sn_grad   the 1/3-resolution cloud, depth-change map, central-difference gradients
//   k_sn_serial    the order-dependent parts, one lane each on separate waves: the two-pass chamfer distance map and the
//                  integral images (double sums in PCL's recurrence order) of the six gradient channels + two finite counts
//   k_sn_normals   normals at the odd grid positions from four-corner integral look-ups
#include "hvo_internal.hpp"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <vector>

#define PT_MAXPL 64
#define PT_TABCAP (1 << 18)                 // voxel-table cells per frame (all planes together)

struct PcPlane {                             // per plane, device
    unsigned bmin[3], bmax[3];               // ordered-int encodings of the float bounding box
    int npix;
    int minb[3], divb[3];
    int tab_off, cells;
    float coef[4];
    int first, npts, gate_ok, valid, ninl;
};

static __device__ __forceinline__ unsigned f2ord(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
static __device__ __forceinline__ float ord2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// the float point of pixel (i, j): PlaneDetection::readDepthImage's double point (PlaneExtractor.cpp:42-56) cast to float (Frame.cc:2116-2119)
static __device__ __forceinline__ void pixel_point(const uint16_t *depth, int pitch, int i, int j, float fx, float fy, float cx, float cy, float dfac, float p[3])
{
    const double z = (double)depth[(size_t)i * pitch + j] * (double)dfac;
    p[0] = (float)(((double)j - (double)cx) * z / (double)fx); p[1] = (float)(((double)i - (double)cy) * z / (double)fy); p[2] = (float)z;
}

// (labels: int32 at the ABI, int8 where the plane stage left them in HBM; npl_ptr: the plane count when it only exists on the device)
template <class LT>
__global__ __launch_bounds__(256) void k_pc_bbox(const uint16_t *__restrict__ depth, int pitch, int w, int h, const LT *__restrict__ labels, const int *__restrict__ npl_ptr, int npl_fixed,
                                                 float fx, float fy, float cx, float cy, float dfac, PcPlane *__restrict__ P)
{
    __shared__ unsigned smn[PT_MAXPL][3], smx[PT_MAXPL][3]; __shared__ int scnt[PT_MAXPL];
    const int npl = npl_ptr ? min(*npl_ptr, npl_fixed) : npl_fixed;
    const int tid = threadIdx.x;
    if (tid < PT_MAXPL) { for (int k = 0; k < 3; k++) { smn[tid][k] = 0xFFFFFFFFu; smx[tid][k] = 0u; } scnt[tid] = 0; }
    __syncthreads();
    const int npix = w * h;
    for (int px = blockIdx.x * 256 + tid; px < npix; px += gridDim.x * 256) {
        const int l = (int)labels[px];
        if (l < 0 || l >= npl) continue;
        const int i = px / w, j = px - i * w;
        float p[3]; pixel_point(depth, pitch, i, j, fx, fy, cx, cy, dfac, p);
        for (int k = 0; k < 3; k++) { const unsigned o = f2ord(p[k]); atomicMin(&smn[l][k], o); atomicMax(&smx[l][k], o); }
        atomicAdd(&scnt[l], 1);
    }
    __syncthreads();
    if (tid < npl && scnt[tid]) {
        for (int k = 0; k < 3; k++) { atomicMin(&P[tid].bmin[k], smn[tid][k]); atomicMax(&P[tid].bmax[k], smx[tid][k]); }
        atomicAdd(&P[tid].npix, scnt[tid]);
    }
}

__global__ void k_pc_init(PcPlane *__restrict__ P, int *__restrict__ misc)
{
    const int t = threadIdx.x;
    if (t < PT_MAXPL) { PcPlane q; memset(&q, 0, sizeof(q)); for (int k = 0; k < 3; k++) { q.bmin[k] = 0xFFFFFFFFu; q.bmax[k] = 0u; } P[t] = q; }
    if (t < 2) misc[t] = 0;
}

__global__ void k_pc_setup(PcPlane *__restrict__ P, const hvo_plane *__restrict__ planes, const int *__restrict__ npl_ptr, int npl_fixed, int *__restrict__ flags)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int npl = npl_ptr ? min(*npl_ptr, npl_fixed) : npl_fixed;
    int off = 0;
    const float inv_leaf = __fdiv_rn(1.0f, 0.1f);
    for (int pl = 0; pl < npl; pl++) {
        PcPlane &q = P[pl];
        const double nx = planes[pl].normal[0], ny = planes[pl].normal[1], nz = planes[pl].normal[2];
        q.coef[0] = (float)nx; q.coef[1] = (float)ny; q.coef[2] = (float)nz;
        q.coef[3] = (float)-(nx * planes[pl].center[0] + ny * planes[pl].center[1] + nz * planes[pl].center[2]);
        q.tab_off = off; q.cells = 0; q.first = 0; q.npts = 0; q.gate_ok = 0; q.valid = 0; q.ninl = 0;
        if (q.npix > 0) {
            long long cells = 1;
            for (int k = 0; k < 3; k++) {
                q.minb[k] = (int)floorf(__fmul_rn(ord2f(q.bmin[k]), inv_leaf));
                const int maxb = (int)floorf(__fmul_rn(ord2f(q.bmax[k]), inv_leaf));
                q.divb[k] = maxb - q.minb[k] + 1;
                cells *= q.divb[k];
            }
            if (off + cells > PT_TABCAP) { *flags |= 1; q.npix = 0; }
            else { q.cells = (int)cells; off += (int)cells; }
        }
    }
}

struct PcVox { unsigned long long s[3]; unsigned n; unsigned pad; };

template <class LT>
__global__ __launch_bounds__(256) void k_pc_accum(const uint16_t *__restrict__ depth, int pitch, int w, int h, const LT *__restrict__ labels, const int *__restrict__ npl_ptr, int npl_fixed,
                                                  float fx, float fy, float cx, float cy, float dfac, const PcPlane *__restrict__ P, PcVox *__restrict__ tab)
{
    const int npl = npl_ptr ? min(*npl_ptr, npl_fixed) : npl_fixed;
    const int npix = w * h;
    const float inv_leaf = __fdiv_rn(1.0f, 0.1f);
    for (int px = blockIdx.x * 256 + threadIdx.x; px < npix; px += gridDim.x * 256) {
        const int l = (int)labels[px];
        if (l < 0 || l >= npl) continue;
        const PcPlane &q = P[l];
        if (q.cells == 0) continue;
        const int i = px / w, j = px - i * w;
        float p[3]; pixel_point(depth, pitch, i, j, fx, fy, cx, cy, dfac, p);
        const int i0 = (int)__fsub_rn(floorf(__fmul_rn(p[0], inv_leaf)), (float)q.minb[0]), i1 = (int)__fsub_rn(floorf(__fmul_rn(p[1], inv_leaf)), (float)q.minb[1]),
                  i2 = (int)__fsub_rn(floorf(__fmul_rn(p[2], inv_leaf)), (float)q.minb[2]);
        PcVox *v = tab + q.tab_off + (size_t)i0 + (size_t)i1 * q.divb[0] + (size_t)i2 * q.divb[0] * q.divb[1];
        for (int k = 0; k < 3; k++) atomicAdd(&v->s[k], (unsigned long long)__double2ll_rn((double)p[k] * 16777216.0));
        atomicAdd(&v->n, 1u);
    }
}

__global__ __launch_bounds__(256) void k_pc_count(PcPlane *__restrict__ P, const PcVox *__restrict__ tab)
{
    __shared__ int s_cnt;
    PcPlane &q = P[blockIdx.x];
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int c = threadIdx.x; c < q.cells; c += 256) mine += tab[q.tab_off + c].n != 0;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) q.npts = s_cnt;
}

__global__ void k_pc_prefix(PcPlane *__restrict__ P, const int *__restrict__ npl_ptr, int npl_fixed, int *__restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int npl = npl_ptr ? min(*npl_ptr, npl_fixed) : npl_fixed;
    int t = 0;
    for (int pl = 0; pl < npl; pl++) { P[pl].first = t; t += P[pl].npts; }
    *total = t;
}

__global__ __launch_bounds__(256) void k_pc_emit(PcPlane *__restrict__ P, const PcVox *__restrict__ tab, float *__restrict__ cloud, int cap, double dist_th)
{
    __shared__ int wsum[4]; __shared__ int s_bad;
    PcPlane &q = P[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    int base = 0;
    for (int c0 = 0; c0 < q.cells; c0 += 256) {
        const int c = c0 + tid;
        PcVox v; v.n = 0;
        if (c < q.cells) v = tab[q.tab_off + c];
        const bool ne = v.n != 0;
        const unsigned long long m = __ballot(ne);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int off = base + __popcll(m & ((1ull << lane) - 1));
        for (int i = 0; i < wv; i++) off += wsum[i];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (ne) {
            const double den = (double)v.n * 16777216.0;
            float qx[3];
            for (int k = 0; k < 3; k++) qx[k] = (float)((double)(long long)v.s[k] / den);
            const int o = q.first + off;
            if (o < cap) { cloud[3 * (size_t)o] = qx[0]; cloud[3 * (size_t)o + 1] = qx[1]; cloud[3 * (size_t)o + 2] = qx[2]; }
            // MaxPointDistanceFromPlane's gate (Frame.cc:2226-2234)
            const float dd = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(q.coef[0], qx[0]), __fmul_rn(q.coef[1], qx[1])), __fmul_rn(q.coef[2], qx[2])), q.coef[3]);
            if (fabs((double)dd) > dist_th) s_bad = 1;
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) q.gate_ok = (q.npts > 0 && !s_bad) ? 1 : 0;
}

// ---- pcl::eigen33 (smallest eigenpair, float): closed-form roots, oracle/planes_tail.c compute_roots / pcl_eigen33
static __device__ void pt_roots2(float b, float c, float r[3])
{
    r[0] = 0.f;
    float d = (float)((double)__fmul_rn(b, b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    const float sd = sqrtf(d);
    r[2] = __fmul_rn(0.5f, __fadd_rn(b, sd));
    r[1] = __fmul_rn(0.5f, __fsub_rn(b, sd));
}
#define FM(a, b) __fmul_rn(a, b)
#define FA(a, b) __fadd_rn(a, b)
#define FS(a, b) __fsub_rn(a, b)
static __device__ void pt_compute_roots(const float m[3][3], float r[3])
{
    const float c0 = FS(FS(FS(FA(FM(FM(m[0][0], m[1][1]), m[2][2]), FM(FM(FM(2.f, m[0][1]), m[0][2]), m[1][2])), FM(FM(m[0][0], m[1][2]), m[1][2])), FM(FM(m[1][1], m[0][2]), m[0][2])),
                        FM(FM(m[2][2], m[0][1]), m[0][1]));
    const float c1 = FS(FA(FS(FA(FS(FM(m[0][0], m[1][1]), FM(m[0][1], m[0][1])), FM(m[0][0], m[2][2])), FM(m[0][2], m[0][2])), FM(m[1][1], m[2][2])), FM(m[1][2], m[1][2]));
    const float c2 = FA(FA(m[0][0], m[1][1]), m[2][2]);
    if (fabsf(c0) < FLT_EPSILON) { pt_roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = FM(c2, s_inv3);
    float a_over_3 = FM(FS(c1, FM(c2, c2_over_3)), s_inv3);
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = FM(0.5f, FA(c0, FM(c2_over_3, FS(FM(FM(2.f, c2_over_3), c2_over_3), c1))));
    float q = FA(FM(half_b, half_b), FM(FM(a_over_3, a_over_3), a_over_3));
    if (q > 0.f) q = 0.f;
    const float rho = sqrtf(-a_over_3);
    const float theta = FM(atan2f(sqrtf(-q), half_b), s_inv3);
    const float cos_theta = cosf(theta), sin_theta = sinf(theta);
    r[0] = FA(c2_over_3, FM(FM(2.f, rho), cos_theta));
    r[1] = FS(c2_over_3, FM(rho, FA(cos_theta, FM(s_sqrt3, sin_theta))));
    r[2] = FS(c2_over_3, FM(rho, FS(cos_theta, FM(s_sqrt3, sin_theta))));
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) { t = r[1]; r[1] = r[2]; r[2] = t; if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; } }
    if (r[0] <= 0) pt_roots2(c2, c1, r);
}
static __device__ void pt_cross(const float a[3], const float b[3], float o[3])
{
    o[0] = FS(FM(a[1], b[2]), FM(a[2], b[1])); o[1] = FS(FM(a[2], b[0]), FM(a[0], b[2])); o[2] = FS(FM(a[0], b[1]), FM(a[1], b[0]));
}
static __device__ void pt_eigen33(const float mat[3][3], float ev[3])
{
    float scale = 0.f;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (fabsf(mat[i][j]) > scale) scale = fabsf(mat[i][j]);
    if (scale <= FLT_MIN) scale = 1.0f;
    float sm[3][3], r[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sm[i][j] = __fdiv_rn(mat[i][j], scale);
    pt_compute_roots(sm, r);
    sm[0][0] = FS(sm[0][0], r[0]); sm[1][1] = FS(sm[1][1], r[0]); sm[2][2] = FS(sm[2][2], r[0]);
    float v1[3], v2[3], v3[3];
    pt_cross(sm[0], sm[1], v1); pt_cross(sm[0], sm[2], v2); pt_cross(sm[1], sm[2], v3);
    const float l1 = FA(FA(FM(v1[0], v1[0]), FM(v1[1], v1[1])), FM(v1[2], v1[2])), l2 = FA(FA(FM(v2[0], v2[0]), FM(v2[1], v2[1])), FM(v2[2], v2[2])),
                l3 = FA(FA(FM(v3[0], v3[0]), FM(v3[1], v3[1])), FM(v3[2], v3[2]));
    const float *v; float l;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; } else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; } else { v = v3; l = l3; }
    const float s = sqrtf(l);
    ev[0] = __fdiv_rn(v[0], s); ev[1] = __fdiv_rn(v[1], s); ev[2] = __fdiv_rn(v[2], s);
}
static __device__ __forceinline__ float pt_dot4(const float m[4], const float *p)
{
    return FA(FA(FA(FM(m[0], p[0]), FM(m[1], p[1])), FM(m[2], p[2])), FM(m[3], 1.0f));
}

// pcl::SACSegmentation (SACMODEL_PLANE, SAC_RANSAC, optimised) on the plane's voxel cloud; oracle/planes_tail.c orc_sac_plane
__global__ __launch_bounds__(256) void k_pc_refit(PcPlane *__restrict__ P, const float *__restrict__ cloud, int cap, int *__restrict__ shuf_all, double threshold)
{
    __shared__ unsigned mt[624]; __shared__ int mti;
    __shared__ float s_mc[4]; __shared__ int s_ctl[4]; __shared__ int s_cnt;
    PcPlane &q = P[blockIdx.x];
    const int tid = threadIdx.x, n = q.npts;
    if (!q.gate_ok || n < 3 || q.first + n > cap) return;                    // (uniform)
    const float *xyz = cloud + 3 * (size_t)q.first;
    int *shuf = shuf_all + q.first;
    for (int i = tid; i < n; i += 256) shuf[i] = i;
    if (tid == 0) { mt[0] = 12345u; for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i; mti = 624; }
    __syncthreads();
    int iterations = 0, best = -INT_MAX; double k = 1.0;
    const int max_iterations = 50;
    const double log_probability = log(1.0 - 0.99), one_over = 1.0 / (double)n;
    unsigned skipped = 0; const unsigned max_skip = max_iterations * 10;
    float model[4] = { 0, 0, 0, 0 }; bool have = false;
    while (iterations < k && skipped < max_skip) {
        if (tid == 0) {
            int good = 0, sel[3] = { 0, 0, 0 };
            for (unsigned it = 0; it < 1000 && !good; it++) {
                for (int i = 0; i < 3; i++) {
                    if (mti >= 624) {
                        for (int kk = 0; kk < 624; kk++) {
                            const unsigned y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % 624] & 0x7FFFFFFFu);
                            mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
                        }
                        mti = 0;
                    }
                    unsigned y = mt[mti++];
                    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
                    const int j = i + (int)((y >> 1) % (unsigned)(n - i));
                    const int t = shuf[i]; shuf[i] = shuf[j]; shuf[j] = t;
                }
                sel[0] = shuf[0]; sel[1] = shuf[1]; sel[2] = shuf[2];
                const float *p0 = xyz + 3 * sel[0], *p1 = xyz + 3 * sel[1], *p2 = xyz + 3 * sel[2];
                const float d0 = __fdiv_rn(FS(p1[0], p0[0]), FS(p2[0], p0[0])), d1 = __fdiv_rn(FS(p1[1], p0[1]), FS(p2[1], p0[1])), d2 = __fdiv_rn(FS(p1[2], p0[2]), FS(p2[2], p0[2]));
                good = (d0 != d1) || (d2 != d1);
            }
            int st = good ? 1 : 0;                                          // 0: no sample, 1: model, 2: collinear (skipped)
            if (good) {
                const float *p0 = xyz + 3 * sel[0], *p1 = xyz + 3 * sel[1], *p2 = xyz + 3 * sel[2];
                const float a[3] = { FS(p1[0], p0[0]), FS(p1[1], p0[1]), FS(p1[2], p0[2]) }, b[3] = { FS(p2[0], p0[0]), FS(p2[1], p0[1]), FS(p2[2], p0[2]) };
                const float e0 = __fdiv_rn(a[0], b[0]), e1 = __fdiv_rn(a[1], b[1]), e2 = __fdiv_rn(a[2], b[2]);
                if ((e0 == e1) && (e2 == e1)) st = 2;
                else {
                    float mc[4];
                    pt_cross(a, b, mc); mc[3] = 0;
                    const float nrm = sqrtf(FA(FA(FA(FM(mc[0], mc[0]), FM(mc[1], mc[1])), FM(mc[2], mc[2])), FM(mc[3], mc[3])));
                    for (int c = 0; c < 4; c++) mc[c] = __fdiv_rn(mc[c], nrm);
                    mc[3] = FM(-1.f, FA(FA(FA(FM(mc[0], p0[0]), FM(mc[1], p0[1])), FM(mc[2], p0[2])), FM(mc[3], 1.0f)));
                    for (int c = 0; c < 4; c++) s_mc[c] = mc[c];
                }
            }
            s_ctl[0] = st; s_cnt = 0;
        }
        __syncthreads();
        const int st = s_ctl[0];
        if (st == 0) break;
        if (st == 2) { ++skipped; __syncthreads(); continue; }
        const float mc[4] = { s_mc[0], s_mc[1], s_mc[2], s_mc[3] };
        int mine = 0;
        for (int i = tid; i < n; i += 256) mine += fabs((double)pt_dot4(mc, xyz + 3 * i)) < threshold;
        if (mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        const int cnt = s_cnt;
        __syncthreads();
        if (cnt > best) {
            best = cnt; for (int c = 0; c < 4; c++) model[c] = mc[c]; have = true;
            const double wq = (double)best * one_over;
            double p_no = 1.0 - pow(wq, 3.0);
            if (p_no < DBL_EPSILON) p_no = DBL_EPSILON;
            if (p_no > 1.0 - DBL_EPSILON) p_no = 1.0 - DBL_EPSILON;
            k = log_probability / log(p_no);
        }
        ++iterations;
        if (iterations > max_iterations) break;
    }
    if (!have) return;
    // inliers of the best model in index order: float covariance sums (order dependent -> one lane), pcl::eigen33, refined inliers
    if (tid == 0) {
        float accf[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        int ninl = 0;
        for (int i = 0; i < n; i++) {
            const float *p = xyz + 3 * i;
            if (!(fabs((double)pt_dot4(model, p)) < threshold)) continue;
            ninl++;
            accf[0] = FA(accf[0], FM(p[0], p[0])); accf[1] = FA(accf[1], FM(p[0], p[1])); accf[2] = FA(accf[2], FM(p[0], p[2])); accf[3] = FA(accf[3], FM(p[1], p[1]));
            accf[4] = FA(accf[4], FM(p[1], p[2])); accf[5] = FA(accf[5], FM(p[2], p[2])); accf[6] = FA(accf[6], p[0]); accf[7] = FA(accf[7], p[1]); accf[8] = FA(accf[8], p[2]);
        }
        float coef[4] = { model[0], model[1], model[2], model[3] };
        if (ninl > 3) {
            for (int c = 0; c < 9; c++) accf[c] = __fdiv_rn(accf[c], (float)ninl);
            float cov[3][3];
            cov[0][0] = FS(accf[0], FM(accf[6], accf[6])); cov[0][1] = FS(accf[1], FM(accf[6], accf[7])); cov[0][2] = FS(accf[2], FM(accf[6], accf[8]));
            cov[1][1] = FS(accf[3], FM(accf[7], accf[7])); cov[1][2] = FS(accf[4], FM(accf[7], accf[8])); cov[2][2] = FS(accf[5], FM(accf[8], accf[8]));
            cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
            float evec[3];
            pt_eigen33(cov, evec);
            coef[0] = evec[0]; coef[1] = evec[1]; coef[2] = evec[2]; coef[3] = 0;
            coef[3] = FM(-1.f, FA(FA(FA(FM(coef[0], accf[6]), FM(coef[1], accf[7])), FM(coef[2], accf[8])), FM(coef[3], 1.0f)));
        }
        s_ctl[1] = ninl;
        for (int c = 0; c < 4; c++) s_mc[c] = coef[c];
        s_cnt = 0;
    }
    __syncthreads();
    if (s_ctl[1] == 0) return;
    const float coef[4] = { s_mc[0], s_mc[1], s_mc[2], s_mc[3] };
    int mine = 0;
    for (int i = tid; i < n; i += 256) mine += fabs((double)pt_dot4(coef, xyz + 3 * i)) < threshold;
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) {
        q.ninl = s_cnt;
        if (s_cnt > 0) {
            const float oldVal = q.coef[3], newVal = coef[3];
            const bool flip = (newVal < 0 && oldVal > 0) || (newVal > 0 && oldVal < 0);               // Frame.cc:2262-2268
            for (int c = 0; c < 4; c++) q.coef[c] = flip ? -coef[c] : coef[c];
            q.valid = 1;
        }
    }
}

// =================================================================================================== surface normals
struct SnArgs {
    const uint16_t *depth; int pitch, w, h, W, H;
    float fx, fy, cx, cy, dfac;
    float *P; unsigned char *chg; float *dm; float *gx, *gy; double *IX, *IY; unsigned *CX, *CY;
    hvo_surface_normal *out; int cap;
};

__global__ __launch_bounds__(256) void k_sn_cloud(SnArgs a)
{
    const int N = a.W * a.H;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N; idx += gridDim.x * 256) {
        const int r = idx / a.W, c = idx - r * a.W, m = 3 * r, n = 3 * c;
        const float d = FM((float)a.depth[(size_t)m * a.pitch + n], a.dfac);
        a.P[3 * idx + 2] = d; a.P[3 * idx] = __fdiv_rn(FM(FS((float)n, a.cx), d), a.fx); a.P[3 * idx + 1] = __fdiv_rn(FM(FS((float)m, a.cy), d), a.fy);
        a.chg[idx] = 255;
    }
}

__global__ __launch_bounds__(256) void k_sn_grad(SnArgs a)
{
    const int N = a.W * a.H, W = a.W, H = a.H;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N; idx += gridDim.x * 256) {
        const int ri = idx / W, ci = idx - ri * W;
        // depth-change map: marks are zeros, so the order of the marking does not matter
        if (ri < H - 1 && ci < W - 1) {
            const float dz = a.P[3 * idx + 2], dR = a.P[3 * (idx + 1) + 2], dD = a.P[3 * (idx + W) + 2];
            const float lim = FM(FM(0.05f, FA(fabsf(dz), 1.0f)), 2.0f);
            if (fabs((double)FS(dz, dR)) > (double)lim || !isfinite(dz) || !isfinite(dR)) { a.chg[idx] = 0; a.chg[idx + 1] = 0; }
            if (fabs((double)FS(dz, dD)) > (double)lim || !isfinite(dz) || !isfinite(dD)) { a.chg[idx] = 0; a.chg[idx + W] = 0; }
        }
        float g[6] = { 0, 0, 0, 0, 0, 0 };
        if (ri >= 1 && ri < H - 1 && ci >= 1 && ci < W - 1)
            for (int k = 0; k < 3; k++) { g[k] = FS(a.P[3 * (idx + 1) + k], a.P[3 * (idx - 1) + k]); g[3 + k] = FS(a.P[3 * (idx + W) + k], a.P[3 * (idx - W) + k]); }
        for (int k = 0; k < 3; k++) { a.gx[3 * idx + k] = g[k]; a.gy[3 * idx + k] = g[3 + k]; }
    }
}

// Nine independent order-dependent chains, one per wave: wave 0 the two-pass chamfer distance map, waves 1-6 the integral images of the
// six gradient channels (double sums in PCL's recurrence order), waves 7-8 the finite-element counts.  What is sequential in each is ONE
// value carried along a row (the running minimum + 1, the running sum); everything a step needs besides that value -- the previous row,
// the row's own inputs -- is independent of the chain.  So per row the wave's 64 lanes fetch those into the wave's LDS rows, lane 0 walks
// the row out of LDS (a few dependent ALU operations per element instead of a global-memory round trip), and the lanes store the
// finished row.  Same operations in the same order as the one-lane formulation (18 ms per 640x480 frame) at ~0.6 ms.
__global__ __launch_bounds__(576) void k_sn_serial(SnArgs a)
{
    extern __shared__ double sn_lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, W = a.W, H = a.H, N = W * H, IW = W + 1;
    double *Lw = sn_lds + (size_t)wv * 3 * (W + 2);
    if (wv == 0) {
        float *dm = a.dm;                                                 // (slack of W + 2 floats on both sides for the row-wrapping reads)
        float *prv = (float *)Lw, *cur = prv + (W + 2), *av = cur + (W + 2);
        const float big = (float)(W + H);
        for (int i = lane; i < W + 2; i += 64) { dm[i - W - 2] = big; dm[N + i] = big; }
        for (int i = lane; i < N; i += 64) dm[i] = a.chg[i] == 0 ? 0.0f : big;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < W; c += 64) prv[c] = dm[c];
        for (int ri = 1; ri < H; ri++) {                                  // forward: cur[ci] = min(centre, upLeft + 1.4, up + 1, upRight + 1.4, cur[ci-1] + 1)
            float *row = dm + (size_t)ri * W;
            for (int c = lane; c < W; c += 64) cur[c] = row[c];
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < W; c += 64) {
                float v = cur[c];
                if (c >= 1) {
                    const float upLeft = FA(prv[c - 1], 1.4f), up = FA(prv[c], 1.0f), upRight = FA(c + 1 < W ? prv[c + 1] : cur[0], 1.4f);   // prev[W] is this row's first element
                    const float x = upLeft < up ? upLeft : up, m3 = x < upRight ? x : upRight;
                    if (m3 < v) v = m3;
                }
                av[c] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                float v = av[0];
                cur[0] = v;
#pragma unroll 8
                for (int c = 1; c < W; c++) { const float l = FA(v, 1.0f), q = av[c]; v = l < q ? l : q; cur[c] = v; }
            }
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < W; c += 64) row[c] = cur[c];
            float *t = prv; prv = cur; cur = t;
        }
        // backward: cur[ci] = min(centre, lowerLeft + 1.4, lower + 1, lowerRight + 1.4, cur[ci+1] + 1); prv holds row H-1
        for (int ri = H - 2; ri >= 0; ri--) {
            float *row = dm + (size_t)ri * W;
            for (int c = lane; c < W; c += 64) cur[c] = row[c];
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < W; c += 64) {
                float v = cur[c];
                if (c <= W - 2) {
                    const float lowerLeft = FA(c >= 1 ? prv[c - 1] : cur[W - 1], 1.4f), lower = FA(prv[c], 1.0f), lowerRight = FA(prv[c + 1], 1.4f);     // next[-1] is this row's last element
                    const float x = lowerLeft < lower ? lowerLeft : lower, m3 = x < lowerRight ? x : lowerRight;
                    if (m3 < v) v = m3;
                }
                av[c] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                float v = av[W - 1];
                cur[W - 1] = v;
#pragma unroll 8
                for (int c = W - 2; c >= 0; c--) { const float l = FA(v, 1.0f), q = av[c]; v = l < q ? l : q; cur[c] = v; }
            }
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < W; c += 64) row[c] = cur[c];
            float *t = prv; prv = cur; cur = t;
        }
    } else if (wv <= 6) {
        const int im = (wv - 1) / 3, k = (wv - 1) % 3;
        double *I = im ? a.IY : a.IX; const float *g = im ? a.gy : a.gx;
        double *prv = Lw, *cur = Lw + (W + 2), *ev = cur + (W + 2);         // ev[c]: the element's contribution, NaN-free: (finite ? e[k] : "skip")
        for (int c = lane; c <= W; c += 64) { I[3 * c + k] = 0; prv[c] = 0; }
        for (int r = 0; r < H; r++) {
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < W; c += 64) {
                const float *e = g + 3 * ((size_t)r * W + c);
                const bool ok = isfinite(FA(FA(e[0], e[1]), e[2]));
                ev[c] = ok ? (double)e[k] : __longlong_as_double(0x7FF8000000000000ll);      // NaN marks "not added"
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                cur[0] = 0;
                double left = 0, upleft = prv[0];
#pragma unroll 4
                for (int c = 0; c < W; c++) {
                    const double up = prv[c + 1];
                    double v = up + left - upleft;
                    const double e = ev[c];
                    if (e == e) v += e;
                    cur[c + 1] = v;
                    left = v; upleft = up;
                }
            }
            __builtin_amdgcn_wave_barrier();
            double *out = I + (size_t)(r + 1) * IW * 3;
            for (int c = lane; c <= W; c += 64) out[3 * c + k] = cur[c];
            double *t = prv; prv = cur; cur = t;
        }
    } else {
        const int im = wv - 7;
        unsigned *Cn = im ? a.CY : a.CX; const float *g = im ? a.gy : a.gx;
        unsigned *prv = (unsigned *)Lw, *cur = prv + (W + 2), *fv = cur + (W + 2);
        for (int c = lane; c <= W; c += 64) { Cn[c] = 0; prv[c] = 0; }
        for (int r = 0; r < H; r++) {
            __builtin_amdgcn_wave_barrier();
            for (int c = lane; c < W; c += 64) { const float *e = g + 3 * ((size_t)r * W + c); fv[c] = isfinite(FA(FA(e[0], e[1]), e[2])) ? 1u : 0u; }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                cur[0] = 0;
                unsigned run = 0;                                          // cc[c+1] = cp[c+1] + cc[c] - cp[c] + f  (exact: unsigned)
                for (int c = 0; c < W; c++) { run = prv[c + 1] + run - prv[c] + fv[c]; cur[c + 1] = run; }
            }
            __builtin_amdgcn_wave_barrier();
            unsigned *out = Cn + (size_t)(r + 1) * IW;
            for (int c = lane; c <= W; c += 64) out[c] = cur[c];
            unsigned *t = prv; prv = cur; cur = t;
        }
    }
}

__global__ __launch_bounds__(256) void k_sn_normals(SnArgs a)
{
    const int W = a.W, H = a.H, IW = W + 1, ow = W / 2, oh = H / 2;          // odd positions: n = 1, 3, ... < W
    const int nout = oh * ow;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < nout; o += gridDim.x * 256) {
        const int m = 2 * (o / ow) + 1, n = 2 * (o % ow) + 1, idx = m * W + n;
        const float bad = __uint_as_float(0x7FC00000u);
        float nrm[3] = { bad, bad, bad };
        const int border = 10;
        if (m >= border && m < H - border && n >= border && n < W - border && isfinite(a.P[3 * idx + 2])) {
            const float sm = a.dm[idx] < 10.0f ? a.dm[idx] : 10.0f;
            if (sm > 2.0f) {
                const int rw = (int)sm, rw2 = rw / 2, sx0 = n - rw2, sy0 = m - rw2;
                const size_t ul = (size_t)sy0 * IW + sx0, ur = ul + rw, ll = (size_t)(sy0 + rw) * IW + sx0, lr = ll + rw;
                const unsigned cxn = a.CX[lr] + a.CX[ul] - a.CX[ur] - a.CX[ll], cyn = a.CY[lr] + a.CY[ul] - a.CY[ur] - a.CY[ll];
                if (cxn != 0 && cyn != 0) {
                    double GX[3], GY[3];
                    for (int k = 0; k < 3; k++) { GX[k] = a.IX[3 * lr + k] + a.IX[3 * ul + k] - a.IX[3 * ur + k] - a.IX[3 * ll + k]; GY[k] = a.IY[3 * lr + k] + a.IY[3 * ul + k] - a.IY[3 * ur + k] - a.IY[3 * ll + k]; }
                    const double nv[3] = { GY[1] * GX[2] - GY[2] * GX[1], GY[2] * GX[0] - GY[0] * GX[2], GY[0] * GX[1] - GY[1] * GX[0] };
                    const double len = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
                    if (len != 0.0) {
                        const double s = sqrt(len);
                        float fxn = (float)(nv[0] / s), fyn = (float)(nv[1] / s), fzn = (float)(nv[2] / s);
                        const float vx = FS(0.f, a.P[3 * idx]), vy = FS(0.f, a.P[3 * idx + 1]), vz = FS(0.f, a.P[3 * idx + 2]);
                        const float ct = FA(FA(FM(vx, fxn), FM(vy, fyn)), FM(vz, fzn));
                        if (ct < 0) { fxn = FM(fxn, -1.f); fyn = FM(fyn, -1.f); fzn = FM(fzn, -1.f); }
                        nrm[0] = fxn; nrm[1] = fyn; nrm[2] = fzn;
                    }
                }
            }
        }
        if (o < a.cap) {
            hvo_surface_normal &r = a.out[o];
            r.normal[0] = nrm[0]; r.normal[1] = nrm[1]; r.normal[2] = nrm[2];
            r.position[0] = a.P[3 * idx]; r.position[1] = a.P[3 * idx + 1]; r.position[2] = a.P[3 * idx + 2];
            r.frame_x = n * 3; r.frame_y = m * 3;
        }
    }
}
#undef FM
#undef FA
#undef FS

// =================================================================================================== host side
// PcPlane -> the ABI's hvo_plane_cloud records (+ the totals: out_n[0] = voxel points of all planes, out_n[1] = capacity flag)
__global__ void k_pc_export(const PcPlane *__restrict__ P, const int *__restrict__ npl_ptr, int npl_fixed, const int *__restrict__ misc, int cap,
                            hvo_plane_cloud *__restrict__ out, int *__restrict__ out_n)
{
    const int npl = npl_ptr ? min(*npl_ptr, npl_fixed) : npl_fixed;
    const int i = threadIdx.x;
    if (i < PT_MAXPL) {
        hvo_plane_cloud o; memset(&o, 0, sizeof(o));
        if (i < npl) {
            const PcPlane &q = P[i];
            for (int k = 0; k < 4; k++) o.coef[k] = q.coef[k];
            o.valid = q.valid; o.gate_ok = q.gate_ok; o.first = q.first; o.n_points = q.npts; o.n_pixels = q.npix; o.n_inliers = q.ninl;
        }
        out[i] = o;
    }
    if (i == 0) { out_n[0] = misc[1]; out_n[1] = (misc[0] || misc[1] > cap) ? 1 : 0; }
}

size_t pc_scratch_bytes(int cap)
{
    return PT_MAXPL * sizeof(PcPlane) + (size_t)PT_TABCAP * sizeof(PcVox) + (size_t)cap * sizeof(int) + 256;
}

// device-resident form of the per-plane tail: depth, labels (int8 or int32), the planes and their count (d_npl, or npl_fixed when
// null) already in HBM; scratch of pc_scratch_bytes(cap).  d_cloud: cap x 3 floats; d_out: PT_MAXPL records; d_out_n: 2 ints.
template <class LT>
static int pc_enqueue_t(hvo_ctx *ctx, hipStream_t st, const uint16_t *d_depth, int pitch, int w, int h, const LT *d_labels, const hvo_plane *d_planes,
                        const int *d_npl, int npl_fixed, double dist_th, void *scratch, float *d_cloud, int cap, hvo_plane_cloud *d_out, int *d_out_n)
{
    PcPlane *dP = (PcPlane *)scratch; PcVox *tab = (PcVox *)(dP + PT_MAXPL); int *dshuf = (int *)(tab + PT_TABCAP); int *dmisc = dshuf + cap;
    const hvo_params &p = ctx->p;
    const int nb = std::min((w * h + 255) / 256, 1024);
    const int nplb = d_npl ? PT_MAXPL : npl_fixed;              // per-plane kernels: blocks past the count return at once (q.npix == 0 / q.cells == 0)
    HVO_HIP(hipMemsetAsync(tab, 0, (size_t)PT_TABCAP * sizeof(PcVox), st));
    hipLaunchKernelGGL(k_pc_init, dim3(1), dim3(64), 0, st, dP, dmisc);
    hipLaunchKernelGGL(k_pc_bbox<LT>, dim3(nb), dim3(256), 0, st, d_depth, pitch, w, h, d_labels, d_npl, npl_fixed, p.fx, p.fy, p.cx, p.cy, p.depth_map_factor, dP);
    hipLaunchKernelGGL(k_pc_setup, dim3(1), dim3(1), 0, st, dP, d_planes, d_npl, npl_fixed, dmisc);
    hipLaunchKernelGGL(k_pc_accum<LT>, dim3(nb), dim3(256), 0, st, d_depth, pitch, w, h, d_labels, d_npl, npl_fixed, p.fx, p.fy, p.cx, p.cy, p.depth_map_factor, dP, tab);
    if (nplb > 0) hipLaunchKernelGGL(k_pc_count, dim3(nplb), dim3(256), 0, st, dP, tab);
    hipLaunchKernelGGL(k_pc_prefix, dim3(1), dim3(1), 0, st, dP, d_npl, npl_fixed, dmisc + 1);
    if (nplb > 0) {
        hipLaunchKernelGGL(k_pc_emit, dim3(nplb), dim3(256), 0, st, dP, tab, d_cloud, cap, dist_th);
        hipLaunchKernelGGL(k_pc_refit, dim3(nplb), dim3(256), 0, st, dP, d_cloud, cap, dshuf, dist_th);
    }
    hipLaunchKernelGGL(k_pc_export, dim3(1), dim3(64), 0, st, dP, d_npl, npl_fixed, dmisc, cap, d_out, d_out_n);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}
int pc_enqueue(hvo_ctx *ctx, hipStream_t st, const uint16_t *d_depth, int pitch, int w, int h, const int8_t *d_labels8, const hvo_plane *d_planes,
               const int *d_npl, int npl_fixed, double dist_th, void *scratch, float *d_cloud, int cap, hvo_plane_cloud *d_out, int *d_out_n)
{
    return pc_enqueue_t<int8_t>(ctx, st, d_depth, pitch, w, h, d_labels8, d_planes, d_npl, npl_fixed, dist_th, scratch, d_cloud, cap, d_out, d_out_n);
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// host-array form: a thin wrapper over the enqueue function through the context's staging arena (no allocation per call)
extern "C" int hvo_plane_clouds(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride, const int32_t *labels, const hvo_plane *planes, int n_planes,
                                double dist_th, float *cloud_xyz, int cap, hvo_plane_cloud *out, int *n_total)
{
    if (!ctx || !n_total || n_planes < 0 || n_planes > PT_MAXPL) return HVO_ERR_INVALID_ARG;
    *n_total = 0;
    if (n_planes == 0) return HVO_OK;
    if (!depth || !labels || !planes || !cloud_xyz || !out || cap < 1 || w <= 0 || h <= 0 || stride < 2 * w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hipStream_t st = ctx->stream;
    const size_t b_d = al256((size_t)w * h * 2), b_l = al256((size_t)w * h * 4), b_p = al256(PT_MAXPL * sizeof(hvo_plane)), b_c = al256((size_t)cap * 12),
                 b_o = al256(PT_MAXPL * sizeof(hvo_plane_cloud));
    char *a = (char *)hvo_call_arena(ctx, b_d + b_l + b_p + b_c + b_o + 256 + pc_scratch_bytes(cap));
    if (!a) return HVO_ERR_HIP;
    uint16_t *dd = (uint16_t *)a; int *dl = (int *)(a + b_d); hvo_plane *dpl = (hvo_plane *)(a + b_d + b_l); float *dc = (float *)(a + b_d + b_l + b_p);
    hvo_plane_cloud *dout = (hvo_plane_cloud *)(a + b_d + b_l + b_p + b_c); int *dn = (int *)(a + b_d + b_l + b_p + b_c + b_o);
    void *scratch = a + b_d + b_l + b_p + b_c + b_o + 256;
    HVO_HIP(hipMemcpy2DAsync(dd, (size_t)w * 2, depth, stride, (size_t)w * 2, h, hipMemcpyHostToDevice, st));
    HVO_HIP(hipMemcpyAsync(dl, labels, (size_t)w * h * 4, hipMemcpyHostToDevice, st));
    HVO_HIP(hipMemcpyAsync(dpl, planes, (size_t)n_planes * sizeof(hvo_plane), hipMemcpyHostToDevice, st));
    int rc = pc_enqueue_t<int>(ctx, st, dd, w, w, h, dl, dpl, nullptr, n_planes, dist_th, scratch, dc, cap, dout, dn);
    if (rc) return rc;
    int hn[2] = { 0, 0 };
    std::vector<hvo_plane_cloud> ho(PT_MAXPL);
    HVO_HIP(hipMemcpyAsync(ho.data(), dout, PT_MAXPL * sizeof(hvo_plane_cloud), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipMemcpyAsync(hn, dn, sizeof(hn), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipStreamSynchronize(st));
    *n_total = hn[0];
    if (hn[0] > 0) HVO_HIP(hipMemcpy(cloud_xyz, dc, (size_t)std::min(hn[0], cap) * 3 * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < n_planes; i++) out[i] = ho[i];
    return hn[1] ? HVO_ERR_CAPACITY : HVO_OK;
}

static void sn_carve(SnArgs &a, int w, int h, char *base, size_t &bytes)
{
    a.w = w; a.h = h; a.W = (w + 2) / 3; a.H = (h + 2) / 3;
    const size_t N = (size_t)a.W * a.H, IN = (size_t)(a.W + 1) * (a.H + 1);
    size_t o = 0;
    auto take = [&](size_t b) { char *p = base ? base + o : nullptr; o += al256(b); return p; };
    a.P = (float *)take(3 * N * 4); a.chg = (unsigned char *)take(N + a.W + 2); float *dmbase = (float *)take((N + 2 * (size_t)a.W + 4) * 4);
    a.gx = (float *)take(3 * N * 4); a.gy = (float *)take(3 * N * 4); a.IX = (double *)take(3 * IN * 8); a.IY = (double *)take(3 * IN * 8);
    a.CX = (unsigned *)take(IN * 4); a.CY = (unsigned *)take(IN * 4);
    a.dm = dmbase ? dmbase + a.W + 2 : nullptr;
    bytes = o;
}
size_t sn_scratch_bytes(int w, int h) { SnArgs a; memset(&a, 0, sizeof(a)); size_t b; sn_carve(a, w, h, nullptr, b); return b; }
int sn_count(int w, int h) { return (((h + 2) / 3) / 2) * (((w + 2) / 3) / 2); }

// device-resident form: depth in HBM, scratch of sn_scratch_bytes(w, h), d_out with sn_count(w, h) entries
int sn_enqueue(hvo_ctx *ctx, hipStream_t st, const uint16_t *d_depth, int pitch, int w, int h, void *scratch, hvo_surface_normal *d_out)
{
    SnArgs a; memset(&a, 0, sizeof(a));
    size_t b; sn_carve(a, w, h, (char *)scratch, b);
    a.pitch = pitch; a.depth = d_depth;
    a.fx = ctx->p.fx; a.fy = ctx->p.fy; a.cx = ctx->p.cx; a.cy = ctx->p.cy; a.dfac = ctx->p.depth_map_factor;
    const size_t N = (size_t)a.W * a.H;
    const int nout = sn_count(w, h);
    a.out = d_out; a.cap = nout;
    hipLaunchKernelGGL(k_sn_cloud, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_sn_grad, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, a);
    {
        const size_t lds = (size_t)9 * 3 * (a.W + 2) * sizeof(double);
        if (lds > 160 * 1024) return HVO_ERR_UNSUPPORTED;
        if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_sn_serial), lds)) return HVO_ERR_HIP;
        hipLaunchKernelGGL(k_sn_serial, dim3(1), dim3(576), lds, st, a);
    }
    if (nout > 0) hipLaunchKernelGGL(k_sn_normals, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, a);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

extern "C" int hvo_surface_normals(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride, hvo_surface_normal *out, int cap, int *n)
{
    if (!ctx || !n) return HVO_ERR_INVALID_ARG;
    *n = 0;
    if (!depth || !out || cap < 0 || w < 3 || h < 3 || stride < 2 * w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hipStream_t st = ctx->stream;
    const int nout = sn_count(w, h);
    const size_t b_d = al256((size_t)w * h * 2), b_o = al256((size_t)std::max(nout, 1) * sizeof(hvo_surface_normal));
    char *a = (char *)hvo_call_arena(ctx, b_d + b_o + sn_scratch_bytes(w, h));
    if (!a) return HVO_ERR_HIP;
    uint16_t *dd = (uint16_t *)a; hvo_surface_normal *dout = (hvo_surface_normal *)(a + b_d);
    HVO_HIP(hipMemcpy2DAsync(dd, (size_t)w * 2, depth, stride, (size_t)w * 2, h, hipMemcpyHostToDevice, st));
    int rc = sn_enqueue(ctx, st, dd, w, w, h, a + b_d + b_o, dout);
    if (rc) return rc;
    std::vector<hvo_surface_normal> tmp((size_t)std::max(nout, 1));
    HVO_HIP(hipMemcpyAsync(tmp.data(), dout, (size_t)nout * sizeof(hvo_surface_normal), hipMemcpyDeviceToHost, st));
    HVO_HIP(hipStreamSynchronize(st));
    memcpy(out, tmp.data(), (size_t)std::min(nout, cap) * sizeof(hvo_surface_normal));
    *n = nout;
    return nout > cap ? HVO_ERR_CAPACITY : HVO_OK;
}

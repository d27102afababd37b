/*
 * planes_tail.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * The tail of Frame::ComputePlanes after PlaneDetection (reference src/Frame.cc:2110-2212) and Frame::MaxPointDistanceFromPlane
 * (2214-2274), SURVEY.md 8f.3:
 *   per extracted plane: its pixels' points as float (2113-2122), d = -(n.c) as float (2124-2132), pcl::VoxelGrid(0.1 m)
 *   (2134-2140), the distance gate (2226-2234), the pcl::SACSegmentation plane refit (2236-2250) and the sign rule (2252-2268);
 *   then the 1/3-resolution cloud (2159-2174) and pcl::IntegralImageNormalEstimation(AVERAGE_3D_GRADIENT, 0.05, 10) (2176-2189)
 *   sampled at odd grid positions into vSurfaceNormal (2191-2211).
 *
 * Everything PCL does here is un-vendored (PCL >= 1.7, CMakeLists.txt:51) and restated from PCL 1.8's published sources -- ASSUMED:
 *   - pcl::VoxelGrid<PointT>::applyFilter: bounding box, min_b = floor(min * inverse_leaf) (float), voxel index
 *     i + j * div0 + k * div0 * div1, one centroid per non-empty voxel in ascending index order.  PCL sorts (voxel, point) pairs
 *     with the unstable std::sort and sums the voxel's points in float in that order, which no restatement can reproduce;
 *     here the centroid is the EXACT mean (coordinates accumulated as 2^-24 m fixed point in 64-bit integers -- order
 *     independent) rounded to float once: within n * 2^-24 relative of any float summation order.
 *   - pcl::SACSegmentation (SACMODEL_PLANE, SAC_RANSAC, 50 iterations, probability 0.99, optimize coefficients):
 *     RandomSampleConsensus::computeModel, SampleConsensusModel::drawIndexSample with boost::mt19937 seeded 12345 behind
 *     boost::uniform_int<>(0, INT_MAX) (= mt() >> 1), SampleConsensusModelPlane::isSampleGood / computeModelCoefficients /
 *     countWithinDistance / selectWithinDistance / optimizeModelCoefficients, computeMeanAndCovarianceMatrix (float, single pass)
 *     and pcl::eigen33 (closed-form roots, float atan2 / cos / sin).  Float dot products accumulate left to right.
 *   - pcl::IntegralImageNormalEstimation::computeFeature (BORDER_POLICY_IGNORE, no depth-dependent smoothing): depth-change map,
 *     two-pass 1 / 1.4 chamfer distance map (with its row-wrapping reads), IntegralImage2D<float, 3> (double sums in the
 *     recurrence cur[c+1] = prev[c+1] + cur[c] - prev[c] (+ element), finite-element counts), rect size (int)min(distance, 10),
 *     normal = gy x gx / |.|, flipped towards the origin, NaN where it cannot be computed.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <float.h>

/* ---------------------------------------------------------------- boost::mt19937 */
typedef struct { uint32_t s[624]; int i; } mt_t;
static void mt_seed(mt_t *m, uint32_t seed) { m->s[0] = seed; for (int i = 1; i < 624; i++) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i; m->i = 624; }
static uint32_t mt_next(mt_t *m)
{
    if (m->i >= 624) {
        for (int k = 0; k < 624; k++) {
            const uint32_t y = (m->s[k] & 0x80000000u) | (m->s[(k + 1) % 624] & 0x7FFFFFFFu);
            m->s[k] = m->s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
        }
        m->i = 0;
    }
    uint32_t y = m->s[m->i++];
    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
    return y;
}

/* ---------------------------------------------------------------- pcl::eigen33 (smallest eigenpair, float) */
static void roots2(float b, float c, float r[3])
{
    r[0] = 0.f;
    float d = (float)(b * b - 4.0 * c);
    if (d < 0.0) d = 0.0f;
    const float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}
static void compute_roots(const float m[3][3], float r[3])
{
    const float c0 = m[0][0] * m[1][1] * m[2][2] + 2.f * m[0][1] * m[0][2] * m[1][2] - m[0][0] * m[1][2] * m[1][2] - m[1][1] * m[0][2] * m[0][2] - m[2][2] * m[0][1] * m[0][1];
    const float c1 = m[0][0] * m[1][1] - m[0][1] * m[0][1] + m[0][0] * m[2][2] - m[0][2] * m[0][2] + m[1][1] * m[2][2] - m[1][2] * m[1][2];
    const float c2 = m[0][0] + m[1][1] + m[2][2];
    if (fabsf(c0) < FLT_EPSILON) { roots2(c2, c1, r); return; }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.f) a_over_3 = 0.f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.f) q = 0.f;
    const float rho = sqrtf(-a_over_3);
    const float theta = atan2f(sqrtf(-q), half_b) * s_inv3;
    const float cos_theta = cosf(theta), sin_theta = sinf(theta);
    r[0] = c2_over_3 + 2.f * rho * cos_theta;
    r[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    r[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) { t = r[1]; r[1] = r[2]; r[2] = t; if (r[0] >= r[1]) { t = r[0]; r[0] = r[1]; r[1] = t; } }
    if (r[0] <= 0) roots2(c2, c1, r);
}
static void cross3f(const float a[3], const float b[3], float o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
static void pcl_eigen33(const float mat[3][3], float *eigenvalue, float ev[3])
{
    float scale = 0.f;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (fabsf(mat[i][j]) > scale) scale = fabsf(mat[i][j]);
    if (scale <= FLT_MIN) scale = 1.0f;
    float sm[3][3], r[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) sm[i][j] = mat[i][j] / scale;
    compute_roots(sm, r);
    *eigenvalue = r[0] * scale;
    sm[0][0] -= r[0]; sm[1][1] -= r[0]; sm[2][2] -= r[0];
    float v1[3], v2[3], v3[3];
    cross3f(sm[0], sm[1], v1); cross3f(sm[0], sm[2], v2); cross3f(sm[1], sm[2], v3);
    const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2], l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2], l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    const float *v; float l;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; } else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; } else { v = v3; l = l3; }
    const float s = sqrtf(l);
    ev[0] = v[0] / s; ev[1] = v[1] / s; ev[2] = v[2] / s;
}

/* ---------------------------------------------------------------- pcl::SACSegmentation (plane, RANSAC, optimised) */
static float dot4(const float m[4], const float *p) { return m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3] * 1.0f; }

/* returns the number of inliers (0: no model); coef = refined coefficients */
int orc_sac_plane(const float *xyz, int n, double threshold, float coef[4])
{
    coef[0] = coef[1] = coef[2] = coef[3] = 0;
    if (n < 3) return 0;
    mt_t mt; mt_seed(&mt, 12345u);
    int *shuf = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) shuf[i] = i;
    int iterations = 0, best = -INT_MAX; double k = 1.0;
    const int max_iterations = 50;
    const double log_probability = log(1.0 - 0.99), one_over = 1.0 / (double)n;
    unsigned skipped = 0; const unsigned max_skip = max_iterations * 10;
    float model[4] = { 0, 0, 0, 0 }; int have = 0;
    while (iterations < k && skipped < max_skip) {
        int sel[3], good = 0;
        for (unsigned it = 0; it < 1000 && !good; it++) {                 /* getSamples: max_sample_checks_ */
            for (int i = 0; i < 3; i++) {                                /* drawIndexSample */
                const int j = i + (int)((mt_next(&mt) >> 1) % (uint32_t)(n - i));
                const int t = shuf[i]; shuf[i] = shuf[j]; shuf[j] = t;
            }
            sel[0] = shuf[0]; sel[1] = shuf[1]; sel[2] = shuf[2];
            const float *p0 = xyz + 3 * sel[0], *p1 = xyz + 3 * sel[1], *p2 = xyz + 3 * sel[2];
            const float d0 = (p1[0] - p0[0]) / (p2[0] - p0[0]), d1 = (p1[1] - p0[1]) / (p2[1] - p0[1]), d2 = (p1[2] - p0[2]) / (p2[2] - p0[2]);
            good = (d0 != d1) || (d2 != d1);                             /* isSampleGood */
        }
        if (!good) break;
        const float *p0 = xyz + 3 * sel[0], *p1 = xyz + 3 * sel[1], *p2 = xyz + 3 * sel[2];
        const float a[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] }, b[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
        const float e0 = a[0] / b[0], e1 = a[1] / b[1], e2 = a[2] / b[2];
        if ((e0 == e1) && (e2 == e1)) { ++skipped; continue; }           /* computeModelCoefficients: collinear */
        float mc[4];
        mc[0] = a[1] * b[2] - a[2] * b[1]; mc[1] = a[2] * b[0] - a[0] * b[2]; mc[2] = a[0] * b[1] - a[1] * b[0]; mc[3] = 0;
        const float nrm = sqrtf(mc[0] * mc[0] + mc[1] * mc[1] + mc[2] * mc[2] + mc[3] * mc[3]);
        mc[0] /= nrm; mc[1] /= nrm; mc[2] /= nrm; mc[3] /= nrm;
        mc[3] = -1 * (mc[0] * p0[0] + mc[1] * p0[1] + mc[2] * p0[2] + mc[3] * 1.0f);
        int cnt = 0;
        for (int i = 0; i < n; i++) if (fabs(dot4(mc, xyz + 3 * i)) < threshold) cnt++;
        if (cnt > best) {
            best = cnt; memcpy(model, mc, sizeof(mc)); have = 1;
            const double w = (double)best * one_over;
            double p_no = 1.0 - pow(w, 3.0);
            if (p_no < DBL_EPSILON) p_no = DBL_EPSILON;
            if (p_no > 1.0 - DBL_EPSILON) p_no = 1.0 - DBL_EPSILON;
            k = log_probability / log(p_no);
        }
        ++iterations;
        if (iterations > max_iterations) break;
    }
    free(shuf);
    if (!have) return 0;
    /* selectWithinDistance + optimizeModelCoefficients + refined inliers */
    double acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    float accf[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int ninl = 0;
    for (int i = 0; i < n; i++) {
        const float *p = xyz + 3 * i;
        if (!(fabs(dot4(model, p)) < threshold)) continue;
        ninl++;
        accf[0] += p[0] * p[0]; accf[1] += p[0] * p[1]; accf[2] += p[0] * p[2]; accf[3] += p[1] * p[1]; accf[4] += p[1] * p[2]; accf[5] += p[2] * p[2];
        accf[6] += p[0]; accf[7] += p[1]; accf[8] += p[2];
    }
    (void)acc;
    if (ninl == 0) return 0;
    memcpy(coef, model, sizeof(model));
    if (ninl > 3) {
        for (int q = 0; q < 9; q++) accf[q] /= (float)ninl;
        float cov[3][3];
        cov[0][0] = accf[0] - accf[6] * accf[6]; cov[0][1] = accf[1] - accf[6] * accf[7]; cov[0][2] = accf[2] - accf[6] * accf[8];
        cov[1][1] = accf[3] - accf[7] * accf[7]; cov[1][2] = accf[4] - accf[7] * accf[8]; cov[2][2] = accf[5] - accf[8] * accf[8];
        cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
        float ev, evec[3];
        pcl_eigen33(cov, &ev, evec);
        coef[0] = evec[0]; coef[1] = evec[1]; coef[2] = evec[2]; coef[3] = 0;
        coef[3] = -1 * (coef[0] * accf[6] + coef[1] * accf[7] + coef[2] * accf[8] + coef[3] * 1.0f);
    }
    int nref = 0;
    for (int i = 0; i < n; i++) if (fabs(dot4(coef, xyz + 3 * i)) < threshold) nref++;
    return nref;
}

/* ---------------------------------------------------------------- per-plane clouds: gather, VoxelGrid(0.1), gate, refit */
typedef struct { long long sx, sy, sz; int n; } vox_t;

int orc_plane_clouds(const uint16_t *depth, int w, int h, int stride_bytes, float fx, float fy, float cx, float cy, float depth_factor,
                     const int32_t *labels, const orc_plane *planes, int nplanes, double dist_th,
                     float *cloud_xyz, int cap, orc_plane_cloud *out)
{
    int total = 0;
    const float inv_leaf = 1.0f / 0.1f;
    for (int pl = 0; pl < nplanes; pl++) {
        orc_plane_cloud *o = &out[pl];
        memset(o, 0, sizeof(*o));
        o->first = total;
        const double nx = planes[pl].normal[0], ny = planes[pl].normal[1], nz = planes[pl].normal[2];
        const float d = (float)-(nx * planes[pl].center[0] + ny * planes[pl].center[1] + nz * planes[pl].center[2]);
        o->coef[0] = (float)nx; o->coef[1] = (float)ny; o->coef[2] = (float)nz; o->coef[3] = d;
        /* the plane's points, as PlaneDetection::readDepthImage made them (double), cast to float */
        float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        int npts = 0;
        for (int pass = 0; pass < 2; pass++) {
            static vox_t *tab; static size_t tabcap;
            int minb[3], divb[3]; size_t cells = 0;
            if (pass == 1) {
                if (npts == 0) break;
                int maxb[3];
                for (int k = 0; k < 3; k++) { minb[k] = (int)floorf(mn[k] * inv_leaf); maxb[k] = (int)floorf(mx[k] * inv_leaf); divb[k] = maxb[k] - minb[k] + 1; }
                cells = (size_t)divb[0] * divb[1] * divb[2];
                if (cells > tabcap) { free(tab); tabcap = cells * 2; tab = (vox_t *)malloc(sizeof(vox_t) * tabcap); }
                memset(tab, 0, sizeof(vox_t) * cells);
            }
            for (int i = 0; i < h; i++) {
                const uint16_t *row = (const uint16_t *)((const uint8_t *)depth + (size_t)i * stride_bytes);
                for (int j = 0; j < w; j++) {
                    if (labels[i * w + j] != pl) continue;
                    const double z = (double)row[j] * depth_factor;
                    const float p[3] = { (float)(((double)j - cx) * z / fx), (float)(((double)i - cy) * z / fy), (float)z };
                    if (pass == 0) { npts++; for (int k = 0; k < 3; k++) { if (p[k] < mn[k]) mn[k] = p[k]; if (p[k] > mx[k]) mx[k] = p[k]; } }
                    else {
                        const int i0 = (int)(floorf(p[0] * inv_leaf) - (float)minb[0]), i1 = (int)(floorf(p[1] * inv_leaf) - (float)minb[1]), i2 = (int)(floorf(p[2] * inv_leaf) - (float)minb[2]);
                        vox_t *v = &tab[(size_t)i0 + (size_t)i1 * divb[0] + (size_t)i2 * divb[0] * divb[1]];
                        v->sx += llrint((double)p[0] * 16777216.0); v->sy += llrint((double)p[1] * 16777216.0); v->sz += llrint((double)p[2] * 16777216.0); v->n++;
                    }
                }
            }
            if (pass == 1) {
                int valid = 1;
                for (size_t c = 0; c < cells; c++) if (tab[c].n) {
                    const double den = (double)tab[c].n * 16777216.0;
                    const float q[3] = { (float)((double)tab[c].sx / den), (float)((double)tab[c].sy / den), (float)((double)tab[c].sz / den) };
                    if (total < cap) { cloud_xyz[3 * total] = q[0]; cloud_xyz[3 * total + 1] = q[1]; cloud_xyz[3 * total + 2] = q[2]; }
                    total++; o->n_points++;
                    /* MaxPointDistanceFromPlane's gate (Frame.cc:2226-2234): float products and sums, |.| against the double threshold */
                    const double absDis = fabs((double)(o->coef[0] * q[0] + o->coef[1] * q[1] + o->coef[2] * q[2] + o->coef[3]));
                    if (absDis > dist_th) valid = 0;
                }
                o->gate_ok = valid;
                if (valid && o->first + o->n_points <= cap) {
                    float nc[4];
                    const int ninl = orc_sac_plane(cloud_xyz + 3 * (size_t)o->first, o->n_points, dist_th, nc);
                    o->n_inliers = ninl;
                    if (ninl > 0) {
                        const float oldVal = o->coef[3], newVal = nc[3];
                        memcpy(o->coef, nc, sizeof(nc));
                        if ((newVal < 0 && oldVal > 0) || (newVal > 0 && oldVal < 0)) for (int k = 0; k < 4; k++) o->coef[k] = -o->coef[k];
                        o->valid = 1;
                    }
                }
            }
        }
        o->n_pixels = npts;
    }
    return total;
}

/* ---------------------------------------------------------------- surface normals on the 1/3-resolution cloud */
int orc_surface_normals(const uint16_t *depth, int w, int h, int stride_bytes, float fx, float fy, float cx, float cy, float depth_factor,
                        orc_surface_normal *out, int cap)
{
    const int W = (int)ceil(w / 3.0), H = (int)ceil(h / 3.0), N = W * H;
    float *P = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    for (int m = 0, r = 0; m < h; m += 3, r++)
        for (int n = 0, c = 0; n < w; n += 3, c++) {
            const uint16_t raw = *(const uint16_t *)((const uint8_t *)depth + (size_t)m * stride_bytes + 2 * (size_t)n);
            const float d = (float)raw * depth_factor;                      /* imDepth (CV_32F) */
            float *p = P + 3 * ((size_t)r * W + c);
            p[2] = d; p[0] = ((float)n - cx) * p[2] / fx; p[1] = ((float)m - cy) * p[2] / fy;
        }
    /* depth-change map and chamfer distance map */
    unsigned char *chg = (unsigned char *)malloc(N + W + 2); memset(chg, 255, N);
    const float factor = 0.05f;
    for (int ri = 0; ri < H - 1; ri++)
        for (int ci = 0; ci < W - 1; ci++) {
            const int idx = ri * W + ci;
            const float dz = P[3 * idx + 2], dR = P[3 * (idx + 1) + 2], dD = P[3 * (idx + W) + 2];
            const float lim = (factor * (fabsf(dz) + 1.0f) * 2.0f);
            if (fabs(dz - dR) > lim || !isfinite(dz) || !isfinite(dR)) { chg[idx] = 0; chg[idx + 1] = 0; }
            if (fabs(dz - dD) > lim || !isfinite(dz) || !isfinite(dD)) { chg[idx] = 0; chg[idx + W] = 0; }
        }
    float *dm = (float *)malloc(sizeof(float) * ((size_t)N + 2 * W + 4)) + W + 2;          /* slack for the row-wrapping reads */
    for (int i = -W - 2; i < N + W + 2; i++) dm[i] = (float)(W + H);
    for (int i = 0; i < N; i++) dm[i] = chg[i] == 0 ? 0.0f : (float)(W + H);
    for (int ri = 1; ri < H; ri++) {
        float *prev = dm + (size_t)(ri - 1) * W, *cur = dm + (size_t)ri * W;
        for (int ci = 1; ci < W; ci++) {
            const float upLeft = prev[ci - 1] + 1.4f, up = prev[ci] + 1.0f, upRight = prev[ci + 1] + 1.4f, left = cur[ci - 1] + 1.0f, center = cur[ci];
            const float a = upLeft < up ? upLeft : up, b = left < upRight ? left : upRight, mv = a < b ? a : b;
            if (mv < center) cur[ci] = mv;
        }
    }
    for (int ri = H - 2; ri >= 0; ri--) {
        float *next = dm + (size_t)(ri + 1) * W, *cur = dm + (size_t)ri * W;
        for (int ci = W - 2; ci >= 0; ci--) {
            const float lowerLeft = next[ci - 1] + 1.4f, lower = next[ci] + 1.0f, lowerRight = next[ci + 1] + 1.4f, right = cur[ci + 1] + 1.0f, center = cur[ci];
            const float a = lowerLeft < lower ? lowerLeft : lower, b = right < lowerRight ? right : lowerRight, mv = a < b ? a : b;
            if (mv < center) cur[ci] = mv;
        }
    }
    /* gradients (initAverage3DGradientMethod) and their integral images */
    float *gx = (float *)calloc((size_t)N * 3, sizeof(float)), *gy = (float *)calloc((size_t)N * 3, sizeof(float));
    for (int ri = 1; ri < H - 1; ri++)
        for (int ci = 1; ci < W - 1; ci++) {
            const int idx = ri * W + ci;
            for (int k = 0; k < 3; k++) { gx[3 * idx + k] = P[3 * (idx + 1) + k] - P[3 * (idx - 1) + k]; gy[3 * idx + k] = P[3 * (idx + W) + k] - P[3 * (idx - W) + k]; }
        }
    const int IW = W + 1;
    double *IX = (double *)calloc((size_t)IW * (H + 1) * 3, sizeof(double)), *IY = (double *)calloc((size_t)IW * (H + 1) * 3, sizeof(double));
    unsigned *CX = (unsigned *)calloc((size_t)IW * (H + 1), sizeof(unsigned)), *CY = (unsigned *)calloc((size_t)IW * (H + 1), sizeof(unsigned));
    for (int im = 0; im < 2; im++) {
        double *I = im ? IY : IX; unsigned *Cn = im ? CY : CX; const float *g = im ? gy : gx;
        for (int r = 0; r < H; r++) {
            double *prev = I + (size_t)r * IW * 3, *cur = prev + (size_t)IW * 3; unsigned *cp = Cn + (size_t)r * IW, *cc = cp + IW;
            cur[0] = cur[1] = cur[2] = 0; cc[0] = 0;
            for (int c = 0; c < W; c++) {
                for (int k = 0; k < 3; k++) cur[3 * (c + 1) + k] = prev[3 * (c + 1) + k] + cur[3 * c + k] - prev[3 * c + k];
                cc[c + 1] = cp[c + 1] + cc[c] - cp[c];
                const float *e = g + 3 * ((size_t)r * W + c);
                if (isfinite(e[0] + e[1] + e[2])) { for (int k = 0; k < 3; k++) cur[3 * (c + 1) + k] += (double)e[k]; ++cc[c + 1]; }
            }
        }
    }
    /* normals, then the odd grid positions in the reference's order */
    const float bad = NAN;
    const int border = 10;
    int nout = 0;
    for (int m = 0; m < H; m++) {
        if (m % 2 == 0) continue;
        for (int n = 0; n < W; n++) {
            if (n % 2 == 0) continue;
            float nrm[3] = { bad, bad, bad };
            const int idx = m * W + n;
            if (m >= border && m < H - border && n >= border && n < W - border && isfinite(P[3 * idx + 2])) {
                const float sm = dm[idx] < 10.0f ? dm[idx] : 10.0f;
                if (sm > 2.0f) {
                    const int rw = (int)sm, rw2 = rw / 2;
                    const int sx0 = n - rw2, sy0 = m - rw2;
                    const size_t ul = (size_t)sy0 * IW + sx0, ur = ul + rw, ll = (size_t)(sy0 + rw) * IW + sx0, lr = ll + rw;
                    const unsigned cxn = CX[lr] + CX[ul] - CX[ur] - CX[ll], cyn = CY[lr] + CY[ul] - CY[ur] - CY[ll];
                    if (cxn != 0 && cyn != 0) {
                        double GX[3], GY[3];
                        for (int k = 0; k < 3; k++) { GX[k] = IX[3 * lr + k] + IX[3 * ul + k] - IX[3 * ur + k] - IX[3 * ll + k]; GY[k] = IY[3 * lr + k] + IY[3 * ul + k] - IY[3 * ur + k] - IY[3 * ll + k]; }
                        double nv[3] = { GY[1] * GX[2] - GY[2] * GX[1], GY[2] * GX[0] - GY[0] * GX[2], GY[0] * GX[1] - GY[1] * GX[0] };
                        const double len = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2];
                        if (len != 0.0f) {
                            const double s = sqrt(len);
                            float fxn = (float)(nv[0] / s), fyn = (float)(nv[1] / s), fzn = (float)(nv[2] / s);
                            const float vx = 0.f - P[3 * idx], vy = 0.f - P[3 * idx + 1], vz = 0.f - P[3 * idx + 2];      /* flipNormalTowardsViewpoint */
                            const float ct = (vx * fxn + vy * fyn + vz * fzn);
                            if (ct < 0) { fxn *= -1; fyn *= -1; fzn *= -1; }
                            nrm[0] = fxn; nrm[1] = fyn; nrm[2] = fzn;
                        }
                    }
                }
            }
            if (nout < cap) {
                orc_surface_normal *o = &out[nout];
                o->normal[0] = nrm[0]; o->normal[1] = nrm[1]; o->normal[2] = nrm[2];
                o->position[0] = P[3 * idx]; o->position[1] = P[3 * idx + 1]; o->position[2] = P[3 * idx + 2];
                o->frame_x = n * 3; o->frame_y = m * 3;
            }
            nout++;
        }
    }
    free(P); free(chg); free(dm - W - 2); free(gx); free(gy); free(IX); free(IY); free(CX); free(CY);
    return nout;
}

/* ---------------------------------------------------------------- Manhattan::computeNormalsLPVO (src/Manhattan.cpp:237-393)
 * The INTENDED reading: the depth image as CV_32F metres (raw * depth_factor, the
 * conversion of src/Frame.cc:201-203 that the RGB-D constructor does NOT hand over -- as compiled the function reads the raw CV_16U
 * image through at<float>), and removeMatRow / removeMatCol as their USE_CV_RECT branches (the integral image without its zero row and
 * column, so I(v, u) is the inclusive sum over rows <= v, columns <= u).  cv::integral (CV_32F -> CV_64F, ASSUMED OpenCV 3.2): per row a
 * running double sum s += src(y, x); sum(y+1, x+1) = sum(y, x+1) + s.  cv::normalize (NORM_L2): v / sqrt(v.v).
 * normals: n x 3 doubles; depth: n floats; pixel: n x 2 ints (u, v).  Returns n (<= cap entries are written). */
int orc_normals_lpvo(const uint16_t *depth, int w, int h, int stride_bytes, float fx, float fy, float cx, float cy, float depth_factor,
                     double *normals, float *depth_out, int *pixel, int cap)
{
    const int cell_size = 10, norm_density = 15;
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;
    const size_t N = (size_t)w * h;
    float *Z = (float *)malloc(sizeof(float) * N), *V = (float *)calloc(3 * N, sizeof(float));
    float *T[7];                                                 /* 0-2 u tangent, 3-5 v tangent, 6 mask */
    double *I[7];
    for (int k = 0; k < 7; k++) { T[k] = (float *)calloc(N, sizeof(float)); I[k] = (double *)malloc(sizeof(double) * N); }
    for (int v = 0; v < h; v++)
        for (int u = 0; u < w; u++) {
            const uint16_t raw = *(const uint16_t *)((const uint8_t *)depth + (size_t)v * stride_bytes + 2 * (size_t)u);
            const float z = (float)raw * depth_factor;
            Z[(size_t)v * w + u] = z;
            if (z > 0.2f && z < 7.0f) {
                float *p = V + 3 * ((size_t)v * w + u);
                p[0] = ((float)u - cx) * z * invfx; p[1] = ((float)v - cy) * z * invfy; p[2] = z;
            }
        }
#define ZBAD(q) ((q) < 0.2f || (q) > 7.0f)
    for (int u = 1; u < w - 1; u++)
        for (int v = 1; v < h - 1; v++) {
            const size_t i = (size_t)v * w + u;
            if (ZBAD(Z[i]) || ZBAD(Z[i - 1]) || ZBAD(Z[i + 1]) || ZBAD(Z[i - w]) || ZBAD(Z[i + w])) continue;
            T[6][i] = 1.0f;
            for (int k = 0; k < 3; k++) { T[k][i] = V[3 * (i + 1) + k] - V[3 * (i - 1) + k]; T[3 + k][i] = V[3 * (i + w) + k] - V[3 * (i - w) + k]; }
        }
#undef ZBAD
    for (int k = 0; k < 7; k++)                                  /* inclusive integral images, cv::integral's summation order */
        for (int y = 0; y < h; y++) {
            double s = 0;
            for (int x = 0; x < w; x++) { s += (double)T[k][(size_t)y * w + x]; I[k][(size_t)y * w + x] = (y > 0 ? I[k][(size_t)(y - 1) * w + x] : 0.0) + s; }
        }
    int n = 0;
#define BOX(k) (I[k][(size_t)v * w + u] - I[k][(size_t)(v - cell_size) * w + u] - I[k][(size_t)v * w + u - cell_size] + I[k][(size_t)(v - cell_size) * w + u - cell_size])
    for (int v = cell_size; v < h - 1; v += norm_density)
        for (int u = cell_size; u < w - 1; u += norm_density) {
            if (T[6][(size_t)v * w + u] != 1) continue;
            const int numPts = (int)BOX(6);
            const double uv[3] = { BOX(0) / numPts, BOX(1) / numPts, BOX(2) / numPts }, vv[3] = { BOX(3) / numPts, BOX(4) / numPts, BOX(5) / numPts };
            const double nv[3] = { vv[1] * uv[2] - vv[2] * uv[1], vv[2] * uv[0] - vv[0] * uv[2], vv[0] * uv[1] - vv[1] * uv[0] };
            const double len = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
            if (n < cap) {
                const double sc = len > DBL_EPSILON ? 1.0 / len : 0.0;       /* cv::normalize: scale = 1 / norm, 0 when the norm is <= DBL_EPSILON */
                normals[3 * n] = nv[0] * sc; normals[3 * n + 1] = nv[1] * sc; normals[3 * n + 2] = nv[2] * sc;
                depth_out[n] = V[3 * ((size_t)v * w + u) + 2]; pixel[2 * n] = u; pixel[2 * n + 1] = v;
            }
            n++;
        }
#undef BOX
    for (int k = 0; k < 7; k++) { free(T[k]); free(I[k]); }
    free(Z); free(V);
    return n;
}

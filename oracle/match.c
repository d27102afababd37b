/*
 * match.c -- ORACLE (test infrastructure only; see oracle.h).
 *
 * 256-bit Hamming distance and brute-force matching:
 *   ORBmatcher::DescriptorDistance   src/ORBmatcher.cc:1676-1692
 *   LSDmatcher::DescriptorDistance   src/LSDmatcher.cpp:1137-1153
 *   LSDmatcher::matchNNR             src/LSDmatcher.cpp:803-826  (cv::BFMatcher knnMatch k=2, ASSUMED:
 *        exhaustive, ascending distance, ties -> lower train index)
 */
#include "oracle.h"
#include <limits.h>
#include <string.h>

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    /* the SWAR bit-trick of the reference, word by word (8 x 32 bit) */
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

void orc_hamming_matrix(const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d)
{
    for (int i = 0; i < nq; i++)
        for (int j = 0; j < nt; j++)
            d[(size_t)i * nt + j] = (uint16_t)orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
}

void orc_hamming_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2)
{
    for (int i = 0; i < nq; i++) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
            else if (d < b1) { b1 = d; i1 = j; }
        }
        idx2[2 * i] = i0; idx2[2 * i + 1] = i1; dist2[2 * i] = b0; dist2[2 * i + 1] = b1;
    }
}

int orc_match_nnr(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float nnr, int32_t *m12)
{
    int matches = 0;
    for (int i = 0; i < n1; i++) {
        int32_t idx[2], dist[2];
        orc_hamming_knn2(d1 + 32 * (size_t)i, 1, d2, n2, idx, dist);
        m12[i] = -1;
        /* the reference indexes matches_[idx][1] unconditionally: needs n2 >= 2 */
        if (n2 >= 2 && (float)dist[0] < (float)dist[1] * nnr) { m12[i] = idx[0]; matches++; }
    }
    return matches;
}

/* ------------------------------------------------------------------------------------------------
 * ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) core (src/ORBmatcher.cc:1353-1497)
 * with Frame::GetFeaturesInArea (src/Frame.cc:1502-1555), Frame::PosInGrid / AssignFeaturesToGrid
 * (src/Frame.cc:1679-1690, 832-847) and ComputeThreeMaxima (src/ORBmatcher.cc:1630-1673).
 * The caller supplies, per last-frame map point that survived the projection tests (ORBmatcher.cc:
 * 1381-1404): the projected (u,v), the search radius th*scale[octave], the octave band, ur = u - bf*invz
 * (< 0: no stereo check), its descriptor, its key-point angle and whether the map point has
 * observations (then the current-frame feature it claims is skipped by later queries, :1425-1427).
 * ---------------------------------------------------------------------------------------------- */
#include <math.h>
#include <stdlib.h>
#define GRID_COLS 64
#define GRID_ROWS 48

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } kp_t;

int orc_search_by_projection(const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                             const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle,
                             const uint8_t *q_blocks,
                             const void *t_kp_, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                             float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, int check_orientation,
                             int32_t *match_idx, int32_t *match_dist)
{
    const kp_t *t_kp = (const kp_t *)t_kp_;
    const float invW = (float)GRID_COLS / (mnMaxX - mnMinX), invH = (float)GRID_ROWS / (mnMaxY - mnMinY);
    /* AssignFeaturesToGrid */
    int *cell_of = (int *)malloc(sizeof(int) * (nt + 1));
    for (int i = 0; i < nt; i++) {
        int px = (int)round((t_kp[i].x - mnMinX) * invW), py = (int)round((t_kp[i].y - mnMinY) * invH);
        cell_of[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
    }
    uint8_t *occ = (uint8_t *)calloc(nt + 1, 1);
    for (int i = 0; i < nt; i++) occ[i] = t_occupied ? t_occupied[i] : 0;
    int nmatches = 0;
    int *rot_bin = (int *)malloc(sizeof(int) * (nq + 1));
    const float factor = 1.0f / 30;
    for (int i = 0; i < nq; i++) {
        match_idx[i] = -1; match_dist[i] = 256; rot_bin[i] = -1;
        const float x = q_u[i], y = q_v[i], r = q_radius[i];
        const int minLevel = q_min_level[i], maxLevel = q_max_level[i];
        int nMinCellX = (int)floorf((x - mnMinX - r) * invW); if (nMinCellX < 0) nMinCellX = 0;
        if (nMinCellX >= GRID_COLS) continue;
        int nMaxCellX = (int)ceilf((x - mnMinX + r) * invW); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
        if (nMaxCellX < 0) continue;
        int nMinCellY = (int)floorf((y - mnMinY - r) * invH); if (nMinCellY < 0) nMinCellY = 0;
        if (nMinCellY >= GRID_ROWS) continue;
        int nMaxCellY = (int)ceilf((y - mnMinY + r) * invH); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
        if (nMaxCellY < 0) continue;
        const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        int bestDist = 256, bestIdx = -1;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int j = 0; j < nt; j++) {           /* cell content in insertion (= index) order */
                    if (cell_of[j] != ix * GRID_ROWS + iy) continue;
                    if (bCheckLevels) {
                        if (t_kp[j].octave < minLevel) continue;
                        if (maxLevel >= 0 && t_kp[j].octave > maxLevel) continue;
                    }
                    const float distx = t_kp[j].x - x, disty = t_kp[j].y - y;
                    if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
                    if (occ[j]) continue;
                    if (t_uright && t_uright[j] > 0 && q_ur) {
                        const float er = fabsf(q_ur[i] - t_uright[j]);
                        if (er > r) continue;
                    }
                    const int dist = orc_descriptor_distance(q_desc + 32 * (size_t)i, t_desc + 32 * (size_t)j);
                    if (dist < bestDist) { bestDist = dist; bestIdx = j; }
                }
        if (bestIdx >= 0 && bestDist <= th_high) {
            match_idx[i] = bestIdx; match_dist[i] = bestDist; nmatches++;
            if (q_blocks[i]) occ[bestIdx] = 1;
            if (check_orientation) {
                float rot = q_angle[i] - t_kp[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == 30) bin = 0;
                rot_bin[i] = bin;
            }
        }
    }
    if (check_orientation) {
        int hist[30]; for (int b = 0; b < 30; b++) hist[b] = 0;
        for (int i = 0; i < nq; i++) if (rot_bin[i] >= 0) hist[rot_bin[i]]++;
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int b = 0; b < 30; b++) {
            const int s = hist[b];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
            else if (s > max3) { max3 = s; ind3 = b; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) ind3 = -1;
        for (int i = 0; i < nq; i++)
            if (rot_bin[i] >= 0 && rot_bin[i] != ind1 && rot_bin[i] != ind2 && rot_bin[i] != ind3) { match_idx[i] = -1; nmatches--; }
    }
    free(cell_of); free(occ); free(rot_bin);
    return nmatches;
}

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:1940-1961): depth image = raw u16 * depthMapFactor in
 * float (Frame.cc:200-203), indexed with the float key-point coordinates truncated to int. */
void orc_stereo_from_rgbd(const void *kp_, const void *kpun_, int n, const uint16_t *depth, int w, int h, int stride_bytes,
                          float depth_factor, float bf, float *uright, float *zdepth)
{
    const kp_t *kp = (const kp_t *)kp_, *kpu = (const kp_t *)kpun_;
    for (int i = 0; i < n; i++) {
        uright[i] = -1; zdepth[i] = -1;
        const int v = (int)kp[i].y, u = (int)kp[i].x;
        if (u < 0 || v < 0 || u >= w || v >= h) continue;
        const float d = (float)((const uint16_t *)((const uint8_t *)depth + (size_t)v * stride_bytes))[u] * depth_factor;
        if (d > 0 && d < 7.0) { zdepth[i] = d; uright[i] = kpu[i].x - bf / d; }
    }
}

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) core (src/ORBmatcher.cc:45-132):
 * the local-map variant.  One query per map point that is in view (:52-56): projected (u, v) = mTrackProjX/Y,
 * radius = r * scale[level] (:63-70), levels [level-1, level] (:71), ur = mTrackProjXR; best and second-best Hamming
 * distance with their octaves; accepted if best <= TH_HIGH and not (same octave && best > mfNNratio * second)
 * (:117-124).  A feature whose map point has observations is skipped (:89-91): t_occupied, and q_blocks[i] marks
 * a query whose map point has observations (it then blocks the feature it is assigned to). */
int orc_search_by_projection_map(const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                                 const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const uint8_t *q_blocks,
                                 const void *t_kp_, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                 float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                 int32_t *match_idx, int32_t *match_dist)
{
    const kp_t *t_kp = (const kp_t *)t_kp_;
    const float invW = (float)GRID_COLS / (mnMaxX - mnMinX), invH = (float)GRID_ROWS / (mnMaxY - mnMinY);
    int *cell_of = (int *)malloc(sizeof(int) * (nt + 1));
    for (int i = 0; i < nt; i++) {
        int px = (int)round((t_kp[i].x - mnMinX) * invW), py = (int)round((t_kp[i].y - mnMinY) * invH);
        cell_of[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
    }
    uint8_t *occ = (uint8_t *)calloc(nt + 1, 1);
    for (int i = 0; i < nt; i++) occ[i] = t_occupied ? t_occupied[i] : 0;
    int nmatches = 0;
    for (int i = 0; i < nq; i++) {
        match_idx[i] = -1; match_dist[i] = 256;
        const float x = q_u[i], y = q_v[i], r = q_radius[i];
        const int minLevel = q_min_level[i], maxLevel = q_max_level[i];
        int nMinCellX = (int)floorf((x - mnMinX - r) * invW); if (nMinCellX < 0) nMinCellX = 0;
        if (nMinCellX >= GRID_COLS) continue;
        int nMaxCellX = (int)ceilf((x - mnMinX + r) * invW); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
        if (nMaxCellX < 0) continue;
        int nMinCellY = (int)floorf((y - mnMinY - r) * invH); if (nMinCellY < 0) nMinCellY = 0;
        if (nMinCellY >= GRID_ROWS) continue;
        int nMaxCellY = (int)ceilf((y - mnMinY + r) * invH); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
        if (nMaxCellY < 0) continue;
        const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int j = 0; j < nt; j++) {
                    if (cell_of[j] != ix * GRID_ROWS + iy) continue;
                    if (bCheckLevels) {
                        if (t_kp[j].octave < minLevel) continue;
                        if (maxLevel >= 0 && t_kp[j].octave > maxLevel) continue;
                    }
                    const float distx = t_kp[j].x - x, disty = t_kp[j].y - y;
                    if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
                    if (occ[j]) continue;
                    if (t_uright && t_uright[j] > 0 && q_ur) {
                        const float er = fabsf(q_ur[i] - t_uright[j]);
                        if (er > r) continue;
                    }
                    const int dist = orc_descriptor_distance(q_desc + 32 * (size_t)i, t_desc + 32 * (size_t)j);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = t_kp[j].octave; bestIdx = j; }
                    else if (dist < bestDist2) { bestLevel2 = t_kp[j].octave; bestDist2 = dist; }
                }
        if (bestDist <= th_high) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            match_idx[i] = bestIdx; match_dist[i] = bestDist; nmatches++;
            if (q_blocks[i]) occ[bestIdx] = 1;
        }
    }
    free(cell_of); free(occ);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------------
 * LSDmatcher::FrameBFMatch (src/LSDmatcher.cpp:942-966) with lineDescriptorMAD (1110-1135) and the mutual
 * check of LSDmatcher::SearchDouble (902-939).  knnMatch(k=2) per query; nn12 threshold = 0.5 * 1.4826 * median
 * absolute deviation of (d1 - d0); accepted if d1 - d0 > threshold && d0 < TH && d0 < mfNNratio * d1.
 * Medians are order statistics, so the unstable std::sort calls cannot change them.
 * ---------------------------------------------------------------------------------------------- */
static int cmp_float_asc(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
static int cmp_float_desc(const void *a, const void *b) { return -cmp_float_asc(a, b); }

/* the epilogue on a knn-2 table (shared with the HIP library's host side in spirit, not in code) */
static void frame_bf_from_knn2(const int32_t *idx2, const int32_t *dist2, int n1, float TH, float nnratio, int32_t *m12)
{
    float *v = (float *)malloc(sizeof(float) * (n1 + 1));
    for (int i = 0; i < n1; i++) v[i] = (float)dist2[2 * i + 1] - (float)dist2[2 * i];
    qsort(v, n1, sizeof(float), cmp_float_desc);                            /* conpare_descriptor_by_NN12_dist: descending */
    const double nn12_median = (double)v[n1 / 2];
    for (int i = 0; i < n1; i++) v[i] = fabsf((float)((double)((float)dist2[2 * i + 1] - (float)dist2[2 * i]) - nn12_median));
    qsort(v, n1, sizeof(float), cmp_float_asc);
    double nn12_th = 1.4826 * (double)v[n1 / 2];
    nn12_th = nn12_th * 0.5;
    for (int i = 0; i < n1; i++) {
        const float d0 = (float)dist2[2 * i], d1 = (float)dist2[2 * i + 1];
        const double dist_12 = (double)(d1 - d0);
        m12[i] = (dist_12 > nn12_th && d0 < TH && d0 < nnratio * d1) ? idx2[2 * i] : -1;
    }
    free(v);
}

int orc_frame_bf_match(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float TH, float nnratio, int32_t *m12)
{
    for (int i = 0; i < n1; i++) m12[i] = -1;
    if (n1 <= 0 || n2 < 2) return 0;                                         /* knnMatch(k=2) needs two train descriptors */
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 4 * n1), *dist = idx + 2 * n1;
    orc_hamming_knn2(d1, n1, d2, n2, idx, dist);
    frame_bf_from_knn2(idx, dist, n1, TH, nnratio, m12);
    free(idx);
    int m = 0; for (int i = 0; i < n1; i++) m += m12[i] >= 0;
    return m;
}

/* LSDmatcher::SearchDouble core: FrameBFMatch both ways (TH_LOW), keep i -> j only if j -> i */
int orc_search_double(const uint8_t *d1, int n1, const uint8_t *d2, int n2, float TH, float nnratio, int32_t *m12)
{
    for (int i = 0; i < n1; i++) m12[i] = -1;
    if (n1 == 0 || n2 == 0) return 0;
    int32_t *m21 = (int32_t *)malloc(sizeof(int32_t) * (n2 + 1));
    orc_frame_bf_match(d1, n1, d2, n2, TH, nnratio, m12);
    orc_frame_bf_match(d2, n2, d1, n1, TH, nnratio, m21);
    int m = 0;
    for (int i = 0; i < n1; i++) { const int j = m12[i]; if (j >= 0) { if (m21[j] != i) m12[i] = -1; else m++; } }
    free(m21);
    return m;
}



/* ---- the projection prologue of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1364-1405 ----
 * cv::Mat arithmetic of OpenCV 3.2.0 (not vendored; ASSUMED, modules/core/src/matmul.cpp gemm): a plain CV_32F product A(3x3) * b(3x1)
 * [+ c] takes the small-matrix path: t = a0*b0 + a1*b1 + a2*b2 in FLOAT, left to right, then float(t * 1.0 + c * 1.0) in double =
 * float(t + c); a product with a transposed operand (-Rcw.t() * tcw) takes GEMMSingleMul<float, double>: the sum in DOUBLE,
 * float(sum * alpha).  `invzc = 1.0 / z` divides in double and rounds to float; u, v, radius and ur are float expressions. */
static float gemm3_row(const float *a, const float *b, float c) { float t = a[0] * b[0]; t += a[1] * b[1]; t += a[2] * b[2]; return (float)((double)t * 1.0 + (double)c * 1.0); }

/* T = rows 0..2 of mTcw, row-major 3 x 4.  Per query i (a last-frame feature with a map point that is no outlier): world position
 * x3Dw[3 i ..], the feature's octave.  Out: u, v, radius, level band, ur; a point that fails a test (:1385, :1391-1394) gets
 * u = v = 1e30, radius 0 (no grid cell: never searched). */
void orc_project_last(const float *Tcw, const float *Tlw, int n, const float *x3Dw, const int32_t *octave,
                      float fx, float fy, float cx, float cy, float mbf, float mb, int mono, float th, const float *scale_factors,
                      float mnMinX, float mnMinY, float mnMaxX, float mnMaxY,
                      float *q_u, float *q_v, float *q_radius, int32_t *q_min_level, int32_t *q_max_level, float *q_ur, int32_t *fwd_bwd)
{
    const float Rcw[9] = { Tcw[0], Tcw[1], Tcw[2], Tcw[4], Tcw[5], Tcw[6], Tcw[8], Tcw[9], Tcw[10] }, tcw[3] = { Tcw[3], Tcw[7], Tcw[11] };
    const float Rlw[9] = { Tlw[0], Tlw[1], Tlw[2], Tlw[4], Tlw[5], Tlw[6], Tlw[8], Tlw[9], Tlw[10] }, tlw[3] = { Tlw[3], Tlw[7], Tlw[11] };
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) {                     /* twc = -Rcw.t() * tcw */
        double s0 = 0;
        for (int k = 0; k < 3; k++) s0 += (double)Rcw[3 * k + r] * (double)tcw[k];
        twc[r] = (float)(s0 * -1.0);
    }
    for (int r = 0; r < 3; r++) tlc[r] = gemm3_row(Rlw + 3 * r, twc, tlw[r]);
    const int bForward = tlc[2] > mb && !mono, bBackward = -tlc[2] > mb && !mono;
    if (fwd_bwd) { fwd_bwd[0] = bForward; fwd_bwd[1] = bBackward; }
    for (int i = 0; i < n; i++) {
        q_u[i] = q_v[i] = 1e30f; q_radius[i] = 0.f; q_min_level[i] = 0; q_max_level[i] = -1; q_ur[i] = 0.f;
        float x3Dc[3];
        for (int r = 0; r < 3; r++) x3Dc[r] = gemm3_row(Rcw + 3 * r, x3Dw + 3 * i, tcw[r]);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / (double)x3Dc[2]);
        if (invzc < 0) continue;
        const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < mnMinX || u > mnMaxX) continue;
        if (v < mnMinY || v > mnMaxY) continue;
        const int oct = octave[i];
        q_u[i] = u; q_v[i] = v; q_radius[i] = th * scale_factors[oct];
        if (bForward) { q_min_level[i] = oct; q_max_level[i] = -1; }
        else if (bBackward) { q_min_level[i] = 0; q_max_level[i] = oct; }
        else { q_min_level[i] = oct - 1; q_max_level[i] = oct + 1; }
        q_ur[i] = u - mbf * invzc;
    }
}

/* the prologue of SearchByProjection(F, vpMapPoints, th), src/ORBmatcher.cc:55-70 + RadiusByViewingCos (134-140): per map point in view
 * (mbTrackInView, not bad) the window radius r * scale[level] with r = 2.5 (viewCos > 0.998) or 4.0, times th when th != 1.0; levels
 * [level - 1, level] */
void orc_track_windows(int n, const int32_t *level, const float *view_cos, float th, const float *scale_factors,
                       float *q_radius, int32_t *q_min_level, int32_t *q_max_level)
{
    const int bFactor = th != 1.0;
    for (int i = 0; i < n; i++) {
        float r = (double)view_cos[i] > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        q_radius[i] = r * scale_factors[level[i]];
        q_min_level[i] = level[i] - 1; q_max_level[i] = level[i];
    }
}

/* ----------------------------------------------------------------------------------------------
 * The line tracker's own two calls (round 5; SURVEY.md 8(f) widening of a20 / a21 to lines).
 *
 * LSDmatcher::computeAngle2D (src/LSDmatcher.cpp:20-34): |cos| of the angle between two 2-vectors.  The vectors are built in
 * cv::Mat_<double>(1, 2) from FLOAT differences of the key lines' in-octave end points (66-72); cv::Mat::dot on two doubles is
 * a0 b0 then + a1 b1 (ASSUMED OpenCV 3.2 dotProd_64f: no blocking below four elements).
 * ---------------------------------------------------------------------------------------------- */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
static double angle2d_abs_cos(const orc_keyline *a, const orc_keyline *b)
{
    const double ax = (double)(a->eox - a->sox), ay = (double)(a->eoy - a->soy);
    const double bx = (double)(b->eox - b->sox), by = (double)(b->eoy - b->soy);
    double dot = 0.0; dot += ax * bx; dot += ay * by;
    const double ma = sqrt(ax * ax + ay * ay), mb = sqrt(bx * bx + by * by);
    return fabs(dot / (ma * mb));
}

/* LSDmatcher::SearchByGeomNApearance(CurrentFrame, LastFrame, desc_th, matches_12) (src/LSDmatcher.cpp:36-108):
 * match(Last.mLdesc, Cur.mLdesc) = matchNNR (42, 803-826), then per pair, in this order: a last-frame line without a map line is
 * passed over (57: its entry of matches_12 is LEFT as matchNNR set it), so is a pair whose current line has startPointX == 0 (63);
 * the angle gate |cos| >= cos(20 deg) (75-81) and the position gate -- start OR end point within a tenth of the image bounds
 * in both axes (94-98) -- reset the entry to -1; what remains is accepted (CurrentFrame.mvpMapLines[i2] = LastFrame.mvpMapLines[i1]).
 * matches12: n_last entries; accepted: n_last flags; returns lmatches. */
int orc_lines_geom_match(const uint8_t *d_last, const orc_keyline *kl_last, const uint8_t *last_has_mapline, int n_last,
                         const uint8_t *d_cur, const orc_keyline *kl_cur, int n_cur, float desc_th, const float *bounds4,
                         int32_t *matches12, uint8_t *accepted)
{
    for (int i = 0; i < n_last; i++) { matches12[i] = -1; accepted[i] = 0; }
    if (n_last < 1) return 0;
    if (n_cur >= 2) orc_match_nnr(d_last, n_last, d_cur, n_cur, desc_th, matches12);       /* knnMatch(k = 2) needs two train rows (as hvo_match_nnr) */
    const double deltaWidth = (bounds4[1] - bounds4[0]) * 0.1, deltaHeight = (bounds4[3] - bounds4[2]) * 0.1;
    const double th_angle = 20.0, th_rad = th_angle / 180.0 * M_PI, cos_th_angle = cos(th_rad);
    int lmatches = 0;
    for (int i1 = 0; i1 < n_last; ++i1) {
        if (last_has_mapline && !last_has_mapline[i1]) continue;
        const int i2 = matches12[i1];
        if (i2 < 0) continue;
        if (kl_cur[i2].sx == 0) continue;
        const double angle = angle2d_abs_cos(&kl_cur[i2], &kl_last[i1]);
        if (angle < cos_th_angle) { matches12[i1] = -1; continue; }
        const orc_keyline *c = &kl_cur[i2], *l = &kl_last[i1];
        if ((fabs(c->sox - l->sox) > deltaWidth || fabs(c->soy - l->soy) > deltaHeight) && (fabs(c->eox - l->eox) > deltaWidth || fabs(c->eoy - l->eoy) > deltaHeight)) { matches12[i1] = -1; continue; }
        accepted[i1] = 1; ++lmatches;
    }
    return lmatches;
}

/* LSDmatcher::SearchByProjection(CurrentFrame, LastFrame, th) core (src/LSDmatcher.cpp:561-662) over
 * Frame::GetFeaturesInAreaForLine (src/Frame.cc:1557-1627).  One query per last-frame line whose map line is in the current frustum:
 * q_xyxy = (mTrackProjX1, Y1, X2, Y2), q_kl = LastFrame.mvKeylinesUn[i], q_desc = pML->GetDescriptor(), q_blocks != 0 when the map line has
 * observations (the current line it claims is then passed over by later queries, 607-609).  t_*: the current frame's key lines, line
 * functions (mvKeyLineFunctions), descriptors, lines already holding an observed map line, and its line grid (mGridForLine as CSR,
 * cell = ix * 48 + iy).  GetFeaturesInAreaForLine as written: float arithmetic throughout, the three sample points start / middle /
 * end with (x1 + x2) / 2.0 rounded back to float, the window of each in grid cells, cells ix-major, a cell's lines in insertion
 * order, a line kept at its FIRST visit that passes |cos| >= 0.96 against the query's direction and |Lfunc . (x, y, 1)| < r for that
 * sample point (the level arguments are not looked at).  Then, in that order: occupancy, the 10-degree orientation gate on the
 * in-octave end points, the Hamming distance, the length ratio min / max >= 0.75, first strict minimum; accepted if <= 95.
 * (The reference declares a rotation histogram and never fills it.)  Returns nmatches. */
int orc_search_lines_by_projection(int nq, const float *q_xyxy, const orc_keyline *q_kl, const uint8_t *q_desc, const uint8_t *q_blocks,
                                   const orc_keyline *t_kl, const double *t_linefn, const uint8_t *t_desc, const uint8_t *t_occupied, int nt,
                                   const int32_t *cell_start, const int32_t *cell_items, const float *bounds4, float th,
                                   int32_t *match_idx, int32_t *match_dist)
{
    const float mnMinX = bounds4[0], mnMaxX = bounds4[1], mnMinY = bounds4[2], mnMaxY = bounds4[3];
    const float invW = (float)GRID_COLS / (mnMaxX - mnMinX), invH = (float)GRID_ROWS / (mnMaxY - mnMinY);
    const double cos_th_angle = cos(10.0 / 180.0 * M_PI);
    const float TH = 0.96f;
    uint8_t *occ = (uint8_t *)calloc(nt + 1, 1), *seen = (uint8_t *)malloc(nt + 1);
    int *vind = (int *)malloc(sizeof(int) * (nt + 1));
    for (int i = 0; i < nt; i++) occ[i] = t_occupied ? t_occupied[i] : 0;
    int nmatches = 0;
    for (int q = 0; q < nq; q++) {
        match_idx[q] = -1; match_dist[q] = 256;
        const float x1 = q_xyxy[4 * q], y1 = q_xyxy[4 * q + 1], x2 = q_xyxy[4 * q + 2], y2 = q_xyxy[4 * q + 3], r = th;
        const float x[3] = { x1, (float)((x1 + x2) / 2.0), x2 }, y[3] = { y1, (float)((y1 + y2) / 2.0), y2 };
        float delta1x = x1 - x2, delta1y = y1 - y2;
        const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
        delta1x /= norm_delta1; delta1y /= norm_delta1;
        int nv = 0;
        memset(seen, 0, nt + 1);
        for (int i = 0; i < 3; i++) {
            int nMinCellX = (int)floorf((x[i] - mnMinX - r) * invW); if (nMinCellX < 0) nMinCellX = 0;
            if (nMinCellX >= GRID_COLS) continue;
            int nMaxCellX = (int)ceilf((x[i] - mnMinX + r) * invW); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
            if (nMaxCellX < 0) continue;
            int nMinCellY = (int)floorf((y[i] - mnMinY - r) * invH); if (nMinCellY < 0) nMinCellY = 0;
            if (nMinCellY >= GRID_ROWS) continue;
            int nMaxCellY = (int)ceilf((y[i] - mnMinY + r) * invH); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
            if (nMaxCellY < 0) continue;
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                    const int c = ix * GRID_ROWS + iy;
                    for (int k = cell_start[c]; k < cell_start[c + 1]; k++) {
                        const int j = cell_items[k];
                        if (seen[j]) continue;
                        float delta2x = t_kl[j].sx - t_kl[j].ex, delta2y = t_kl[j].sy - t_kl[j].ey;
                        const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
                        delta2x /= norm_delta2; delta2y /= norm_delta2;
                        const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
                        if (CosSita < TH) continue;
                        const float dist = (float)(t_linefn[3 * j] * (double)x[i] + t_linefn[3 * j + 1] * (double)y[i] + t_linefn[3 * j + 2]);
                        if (fabs(dist) < r) { vind[nv++] = j; seen[j] = 1; }
                    }
                }
        }
        if (nv == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int v = 0; v < nv; v++) {
            const int i2 = vind[v];
            if (occ[i2]) continue;
            if (angle2d_abs_cos(&t_kl[i2], &q_kl[q]) < cos_th_angle) continue;
            const int dist = orc_descriptor_distance(q_desc + 32 * (size_t)q, t_desc + 32 * (size_t)i2);
            const float mx = q_kl[q].length > t_kl[i2].length ? q_kl[q].length : t_kl[i2].length;       /* std::max(a, b): a < b ? b : a */
            const float mn = t_kl[i2].length < q_kl[q].length ? t_kl[i2].length : q_kl[q].length;       /* std::min(a, b): b < a ? b : a */
            if (mn / mx < 0.75) continue;
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= 95) {
            match_idx[q] = bestIdx2; match_dist[q] = bestDist; nmatches++;
            if (q_blocks && q_blocks[q]) occ[bestIdx2] = 1;
        }
    }
    free(occ); free(seen); free(vind);
    return nmatches;
}

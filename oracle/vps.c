/*
 * vps.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * The vanishing-point clustering of the key lines that the Frame constructor runs on every frame (reference
 * src/Frame.cc:330-337):
 *   Frame::getVPHypVia2Lines   src/Frame.cc:442-545   line functions, lengths, orientations; 105 x 360 hypotheses
 *   Frame::getSphereGrids      src/Frame.cc:546-650   90 x 360 sphere grid of pairwise intersections + 3x3 smoothing
 *   Frame::getBestVpsHyp       src/Frame.cc:651-707   score of every hypothesis, first maximum
 *   Frame::line2Vps            src/Frame.cc:708-778   cluster of every line (0..2, 3 = none), isStructLine
 *
 * Determinism rule: the reference draws the line pairs from glibc rand(), seeded with the time
 * (src/Frame.cc:476).  Policy: an explicit 32-bit seed; hypothesis group i (one pair of lines, 360 hypotheses) draws from
 * its own xorshift32 stream seeded from (seed, i); a draw is (state & 0x7fffffff) % num like rand() % num; a pair whose
 * intersection has z == 0 is drawn again from the same stream (the reference's "i--; continue").
 * ASSUMED: abs() of a double (src/Frame.cc:611) is std::abs(double); Point2f differences are float subtractions before
 * they widen to double (442-466, 722-731); Frame::fx, fy, cx, cy are floats promoted to double.
 * libm: sin, cos, atan, acos, atan2 are the host's here and the device's in the HIP path; they may differ in the last
 * bit, so the continuous outputs are compared with a tolerance and the discrete ones (grid cells, best hypothesis,
 * clusters) exactly except at ties.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef CV_PI
#define CV_PI 3.1415926535897932384626433832795
#endif

static unsigned xs32(unsigned *s) { unsigned x = *s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; *s = x; return x; }
static void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

int orc_vp_iterations(void)
{
    const double noiseRatio = 0.5, p = 1.0 / 3.0 * pow(1.0 - noiseRatio, 2), confEfficience = 0.9999;
    return (int)(log(1 - confEfficience) / log(1.0 - p));
}

/* para (n x 3), length (n), ori (n): src/Frame.cc:454-473 */
void orc_vp_line_params(const orc_keyline *kl, int n, double *para, double *length, double *ori)
{
    for (int i = 0; i < n; i++) {
        const double p1[3] = { kl[i].sx, kl[i].sy, 1.0 }, p2[3] = { kl[i].ex, kl[i].ey, 1.0 };
        cross3(p1, p2, para + 3 * i);
        const double dx = (double)(kl[i].ex - kl[i].sx), dy = (double)(kl[i].ey - kl[i].sy);
        length[i] = sqrt(dx * dx + dy * dy);
        double o = atan2(dy, dx);
        if (o < 0) o += CV_PI;
        ori[i] = o;
    }
}

#define ORC_VP_REDRAWS 64
/* the 360 hypotheses of group i: hyp[(j * 3 + v) * 3 + c]; returns the pair drawn */
static void vp_group(const double *para, int n, double fx, double cx, double cy, unsigned seed, int i, double *hyp, int *pair)
{
    unsigned rs = seed ^ (0x9E3779B9u * (unsigned)(i + 1)); if (rs == 0) rs = 0x6D2B79F5u;
    const int numVp2 = 360; const double stepVp2 = 2.0 * CV_PI / numVp2;
    double vp1[3];
    /* The reference redraws without bound (`i--; continue`, src/Frame.cc:487-491) and never ends when every pair of lines meets at
     * infinity (all lines parallel in the image).  Bounded here and in k_vp_hyp: after ORC_VP_REDRAWS draws the group gives up and its
     * 360 hypotheses are all zero, which score 0 (vp_score skips z == 0) and can never be the best one unless every group gave up. */
    int tries = 0;
    for (;; tries++) {
        if (tries >= ORC_VP_REDRAWS) { memset(hyp, 0, sizeof(double) * 9 * numVp2); pair[0] = pair[1] = -1; return; }
        const int idx1 = (int)((xs32(&rs) & 0x7fffffffu) % (unsigned)n);
        int idx2 = (int)((xs32(&rs) & 0x7fffffffu) % (unsigned)n);
        while (idx2 == idx1) idx2 = (int)((xs32(&rs) & 0x7fffffffu) % (unsigned)n);
        double v[3];
        cross3(para + 3 * idx1, para + 3 * idx2, v);
        if (v[2] == 0) continue;
        vp1[0] = v[0] / v[2] - cx; vp1[1] = v[1] / v[2] - cy; vp1[2] = fx;
        pair[0] = idx1; pair[1] = idx2;
        break;
    }
    if (vp1[2] == 0) vp1[2] = 0.0011;
    double N = sqrt(vp1[0] * vp1[0] + vp1[1] * vp1[1] + vp1[2] * vp1[2]);
    { const double s = 1.0 / N; vp1[0] *= s; vp1[1] *= s; vp1[2] *= s; }
    for (int j = 0; j < numVp2; j++) {
        const double lambda = j * stepVp2;
        const double k1 = vp1[0] * sin(lambda) + vp1[1] * cos(lambda), k2 = vp1[2];
        const double phi = atan(-k2 / k1);
        double vp2[3] = { sin(phi) * sin(lambda), sin(phi) * cos(lambda), cos(phi) }, vp3[3];
        if (vp2[2] == 0.0) vp2[2] = 0.0011;
        N = sqrt(vp2[0] * vp2[0] + vp2[1] * vp2[1] + vp2[2] * vp2[2]);
        { const double s = 1.0 / N; vp2[0] *= s; vp2[1] *= s; vp2[2] *= s; }
        if (vp2[2] < 0) { vp2[0] *= -1.0; vp2[1] *= -1.0; vp2[2] *= -1.0; }
        cross3(vp1, vp2, vp3);
        if (vp3[2] == 0.0) vp3[2] = 0.0011;
        N = sqrt(vp3[0] * vp3[0] + vp3[1] * vp3[1] + vp3[2] * vp3[2]);
        { const double s = 1.0 / N; vp3[0] *= s; vp3[1] *= s; vp3[2] *= s; }
        if (vp3[2] < 0) { vp3[0] *= -1.0; vp3[1] *= -1.0; vp3[2] *= -1.0; }
        double *h = hyp + (size_t)j * 9;
        memcpy(h, vp1, 24); memcpy(h + 3, vp2, 24); memcpy(h + 6, vp3, 24);
    }
}

/* getSphereGrids (src/Frame.cc:546-650): grid is 90 x 360, row-major; raw = before the 3x3 smoothing (may be NULL) */
void orc_vp_sphere_grid(const double *para, const double *length, const double *ori, int n, double fx, double cx, double cy,
                        double *grid, double *raw)
{
    const double acc = 1.0 / 180.0 * CV_PI;
    const int gridLA = (int)((CV_PI / 2.0) / acc), gridLO = (int)((CV_PI * 2.0) / acc);       /* 90, 360 */
    double *g = (double *)calloc((size_t)gridLA * gridLO, sizeof(double));
    const double tol = 60.0 / 180.0 * CV_PI;
    for (int i = 0; i + 1 < n; i++)
        for (int j = i + 1; j < n; j++) {
            double pt[3];
            cross3(para + 3 * i, para + 3 * j, pt);
            if (pt[2] == 0) continue;
            const double x = pt[0] / pt[2], y = pt[1] / pt[2];
            const double X = x - cx, Y = y - cy, Z = fx, N = sqrt(X * X + Y * Y + Z * Z);
            const double latitude = acos(Z / N), longitude = atan2(X, Y) + CV_PI;
            int LA = (int)(latitude / acc); if (LA >= gridLA) LA = gridLA - 1;
            int LO = (int)(longitude / acc); if (LO >= gridLO) LO = gridLO - 1;
            double dev = fabs(ori[i] - ori[j]);
            dev = fmin(CV_PI - dev, dev);
            if (dev > tol) continue;
            g[LA * gridLO + LO] += sqrt(length[i] * length[j]) * (sin(2.0 * dev) + 0.2);
        }
    if (raw) memcpy(raw, g, sizeof(double) * (size_t)gridLA * gridLO);
    memset(grid, 0, sizeof(double) * (size_t)gridLA * gridLO);
    for (int i = 1; i < gridLA - 1; i++)
        for (int j = 1; j < gridLO - 1; j++) {
            double tot = 0.0;
            for (int m = 0; m < 3; m++) for (int q = 0; q < 3; q++) tot += g[(i - 1 + m) * gridLO + (j - 1 + q)];
            grid[i * gridLO + j] = g[i * gridLO + j] + tot / 9;
        }
    free(g);
}

/* score of one hypothesis (3 x 3 doubles): getBestVpsHyp, src/Frame.cc:660-692 */
static double vp_score(const double *grid, const double *h)
{
    const double oneDegree = 1.0 / 180.0 * CV_PI;
    double s = 0.0;
    for (int j = 0; j < 3; j++) {
        const double *v = h + 3 * j;
        if (v[2] == 0.0) continue;
        const double latitude = acos(v[2]), longitude = atan2(v[0], v[1]) + CV_PI;
        int LA = (int)(latitude / oneDegree); if (LA == 90) LA = 89;
        /* Determinism rule.  The second direction of every hypothesis is built from lambda = j degrees, so in exact arithmetic its
         * longitude IS a whole number of degrees (lambda or lambda +- 180): it sits on a cell boundary, and which of the two cells the
         * reference takes is the rounding noise of its libm's sin / cos / atan / atan2 (observed: 17.00000000000001 and
         * 34.999999999999986).  No two libms agree on that, so a longitude within 1e-6 degree of a whole degree is assigned to
         * that degree's cell on both sides (vps.hip vp_cell_of). */
        const double lo_f = longitude / oneDegree, lo_r = nearbyint(lo_f);
        int LO = fabs(lo_f - lo_r) < 1e-6 ? (int)lo_r : (int)lo_f; if (LO >= 360) LO = 359;
        s += grid[LA * 360 + LO];
    }
    return s;
}

/* line2Vps (src/Frame.cc:708-778): the cluster of every line under the hypothesis vps (3 x 3); 3 = none */
void orc_vp_line2vps(const orc_keyline *kl, int n, float fx_, float fy_, float cx_, float cy_, const double *vps, double th_angle, int32_t *vp_idx)
{
    const double fx = fx_, fy = fy_, cx = cx_, cy = cy_;
    double vx[3], vy[3];
    for (int j = 0; j < 3; j++) { vx[j] = vps[3 * j] * fx / vps[3 * j + 2] + cx; vy[j] = vps[3 * j + 1] * fy / vps[3 * j + 2] + cy; }
    for (int i = 0; i < n; i++) {
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
            angle = fmin(CV_PI - angle, angle);
            if (angle < minAngle) { minAngle = angle; bj = j; }
        }
        vp_idx[i] = minAngle < th_angle ? bj : 3;
    }
}

/* hypothesis `index` (group index / 360, rotation index % 360) as the whole path draws it: 9 doubles */
void orc_vp_hypothesis(const orc_keyline *kl, int n, float fx_, float cx_, float cy_, uint32_t seed, int index, double *hyp9)
{
    double *para = (double *)malloc(sizeof(double) * 5 * (size_t)n), *length = para + 3 * (size_t)n, *ori = length + n;
    double *hyp = (double *)malloc(sizeof(double) * 9 * 360);
    int pair[2];
    orc_vp_line_params(kl, n, para, length, ori);
    vp_group(para, n, fx_, cx_, cy_, seed, index / 360, hyp, pair);
    memcpy(hyp9, hyp + (size_t)(index % 360) * 9, 72);
    free(hyp); free(para);
}

/* The whole path.  vps: 3 x 3 (the best hypothesis); vp_idx: n entries (0..2, 3 = none: local_vp_ids / isStructLine = idx < 3);
 * scores: optional, iterations * 360 entries.  Returns 0, or -1 when n < 2 (the reference skips the path then). */
int orc_vanishing_points(const orc_keyline *kl, int n, float fx_, float fy_, float cx_, float cy_, uint32_t seed, double th_angle,
                         double *vps, int *best_idx, double *best_score, int32_t *vp_idx, double *scores, double *grid_out)
{
    if (n < 2) return -1;
    const double fx = fx_, cx = cx_, cy = cy_;
    double *para = (double *)malloc(sizeof(double) * 5 * (size_t)n), *length = para + 3 * (size_t)n, *ori = length + n;
    orc_vp_line_params(kl, n, para, length, ori);
    const int it = orc_vp_iterations();
    double *hyp = (double *)malloc(sizeof(double) * 9 * 360 * (size_t)it);
    int pair[2];
    for (int i = 0; i < it; i++) vp_group(para, n, fx, cx, cy, seed, i, hyp + (size_t)i * 360 * 9, pair);
    double *grid = (double *)malloc(sizeof(double) * 90 * 360);
    orc_vp_sphere_grid(para, length, ori, n, fx, cx, cy, grid, NULL);
    if (grid_out) memcpy(grid_out, grid, sizeof(double) * 90 * 360);
    int best = 0; double maxLength = 0.0;
    for (int i = 0; i < it * 360; i++) {
        const double s = vp_score(grid, hyp + (size_t)i * 9);
        if (scores) scores[i] = s;
        if (s > maxLength) { maxLength = s; best = i; }
    }
    memcpy(vps, hyp + (size_t)best * 9, 72);
    *best_idx = best; *best_score = maxLength;
    orc_vp_line2vps(kl, n, fx_, fy_, cx_, cy_, vps, th_angle, vp_idx);
    free(grid); free(hyp); free(para);
    return 0;
}

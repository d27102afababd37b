/*
 * line3d.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * Frame::isLineGood (reference src/Frame.cc:1205-1322): the 3-D line of every 2-D key line from the depth image --
 *   samples along the segment, nearest-pixel depth, back-projection                          src/Frame.cc:1214-1270
 *   LINEextractor::compPt3dCov (covariance of a back-projected point, its decomposition)     src/LineExtractor.cpp:44-97 (depthStdDev 31-42)
 *   LINEextractor::extract3dline_mahdist (RANSAC on Mahalanobis point-line distances)        src/LineExtractor.cpp:220-327
 *   LINEextractor::mah_dist3d_pt_line / verify3dLine / computeLine3d_svd / projPt3d2Ln3d      src/LineExtractor.cpp:186-218, 98-160, 162-184; include/LineExtractor.h:227-235
 *   random_unique (partial Fisher-Yates shuffle with rand())                                  include/LineExtractor.h:22-36
 *
 * Determinism rules -- the reference is irreproducible here:
 *   - rand() is glibc's, seeded with the time (src/Frame.cc:476).  Policy: an explicit 32-bit seed; every line draws from its
 *     own xorshift32 stream seeded from (seed, line index), so lines are independent of each other (and of their order);
 *     a draw is (state & 0x7fffffff) % left, like rand() % left.
 *   - cv::SVD of the 3x3 covariance (LineExtractor.cpp:73) and of the n x 3 matrix of centred inliers (:181) live in OpenCV 3.2
 *     (not vendored): ASSUMED any accurate decomposition.  Here: cyclic Jacobi (orc_eig33sym) of the covariance and of the
 *     3x3 scatter matrix; the Mahalanobis distance does not depend on the order or signs of the factors, the refitted
 *     direction's SIGN does (it decides which end point is A and which is B when the refit wins): ASSUMED the Jacobi sign.
 *   - cv::Point_<float> * double rounds to float (OpenCV's operator*), cv::norm of a Point2f / Point3d accumulates in double
 *     in x, y, z order, Mat products are plain dot products in k order: ASSUMED.
 * Quirks kept: the depth bounds test compares row with cols and col with rows (src/Frame.cc:1249), so samples right of
 * column `rows` are dropped; the focal length of BOTH axes in the covariance is K(0,0) (LineExtractor.cpp:51-58).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

#define L3_MAXP 24

typedef struct { double pos[3]; double DU[9]; } rpt_t;

static double depth_std_dev(double d) { const double c1 = 0.00273, c2 = 0.00074, c3 = -0.00058; return c1 * d * d + c2 * d + c3; }

static unsigned xs32(unsigned *s) { unsigned x = *s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; *s = x; return x; }

/* LINEextractor::compPt3dCov (LineExtractor.cpp:44-97) */
static void comp_pt3d_cov(const double p[3], double f, rpt_t *r)
{
    r->pos[0] = p[0]; r->pos[1] = p[1]; r->pos[2] = p[2];
    const double zf = p[2] / f, xz = p[0] / p[2], yz = p[1] / p[2];
    const double sd = depth_std_dev(p[2]), s2 = sd * sd;
    /* cov0 = J0 * diag(1, 1, s2) * J0^T with J0 = [zf 0 xz; 0 zf yz; 0 0 1], as two 3x3 products in k order */
    const double m02 = xz * s2, m12 = yz * s2;
    double K[3][3];
    K[0][0] = zf * zf + m02 * xz; K[0][1] = m02 * yz;            K[0][2] = m02;
    K[1][0] = m12 * xz;           K[1][1] = zf * zf + m12 * yz; K[1][2] = m12;
    K[2][0] = s2 * xz;            K[2][1] = s2 * yz;            K[2][2] = s2;
    /* the decomposition works on the symmetric matrix: the two triangles above agree up to rounding; use the upper one */
    K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
    double w[3], U[3][3];
    orc_eig33sym(K, w, U);                                 /* ascending; cv::SVD orders descending: DU rows in that order */
    for (int i = 0; i < 3; i++) {
        const int c = 2 - i;
        const double inv = 1 / sqrt(w[c]);
        r->DU[3 * i + 0] = inv * U[0][c]; r->DU[3 * i + 1] = inv * U[1][c]; r->DU[3 * i + 2] = inv * U[2][c];
    }
}

/* LINEextractor::mah_dist3d_pt_line (LineExtractor.cpp:186-218) */
static double mah_dist(const rpt_t *pt, const double q1[3], const double q2[3])
{
    const double xa = q1[0], ya = q1[1], za = q1[2], xb = q2[0], yb = q2[1], zb = q2[2];
    const double c1 = pt->DU[0], c2 = pt->DU[1], c3 = pt->DU[2], c4 = pt->DU[3], c5 = pt->DU[4], c6 = pt->DU[5], c7 = pt->DU[6], c8 = pt->DU[7], c9 = pt->DU[8];
    const double x1 = pt->pos[0], x2 = pt->pos[1], x3 = pt->pos[2];
    const double term1 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb)) - (c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term2 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) - (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term3 = ((c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) - (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb))),
                 term4 = (c1 * (x1 - xa) - c1 * (x1 - xb) + c2 * (x2 - ya) - c2 * (x2 - yb) + c3 * (x3 - za) - c3 * (x3 - zb)),
                 term5 = (c4 * (x1 - xa) - c4 * (x1 - xb) + c5 * (x2 - ya) - c5 * (x2 - yb) + c6 * (x3 - za) - c6 * (x3 - zb)),
                 term6 = (c7 * (x1 - xa) - c7 * (x1 - xb) + c8 * (x2 - ya) - c8 * (x2 - yb) + c9 * (x3 - za) - c9 * (x3 - zb));
    return sqrt((term1 * term1 + term2 * term2 + term3 * term3) / (term4 * term4 + term5 * term5 + term6 * term6));
}

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* LINEextractor::verify3dLine (LineExtractor.cpp:98-160) on the points listed in idx */
static int verify_3d_line(const rpt_t *pts, const int *idx, int n, const double A[3], const double B[3])
{
    int cells[10] = { 0 };
    double minv = 100, maxv = -100; int i1 = 0, i2 = 0;
    const double AB[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
    for (int i = 0; i < n; i++) {
        const double *p = pts[idx[i]].pos;
        const double d[3] = { p[0] - A[0], p[1] - A[1], p[2] - A[2] };
        const double v = dot3(d, AB);
        if (v < minv) { minv = v; i1 = i; }
        if (v > maxv) { maxv = v; i2 = i; }
    }
    /* projPt3d2Ln3d(P, mid = (A+B)*0.5, drct = B-A) (LineExtractor.h:227-235) */
    const double mid[3] = { (A[0] + B[0]) * 0.5, (A[1] + B[1]) * 0.5, (A[2] + B[2]) * 0.5 };
    double C[3], D[3];
    for (int e = 0; e < 2; e++) {
        const double *P = pts[idx[e ? i2 : i1]].pos;
        const double Bq[3] = { mid[0] + AB[0], mid[1] + AB[1], mid[2] + AB[2] };
        const double ab[3] = { Bq[0] - mid[0], Bq[1] - mid[1], Bq[2] - mid[2] }, ap[3] = { P[0] - mid[0], P[1] - mid[1], P[2] - mid[2] };
        const double t = dot3(ab, ap) / dot3(ab, ab);
        double *o = e ? D : C;
        o[0] = mid[0] + t * ab[0]; o[1] = mid[1] + t * ab[1]; o[2] = mid[2] + t * ab[2];
    }
    const double DC[3] = { D[0] - C[0], D[1] - C[1], D[2] - C[2] };
    const double cd = sqrt(DC[0] * DC[0] + DC[1] * DC[1] + DC[2] * DC[2]);
    if (cd < 0.0000000001) return 0;
    for (int i = 0; i < n; i++) {
        const double *X = pts[idx[i]].pos;
        const double xc[3] = { X[0] - C[0], X[1] - C[1], X[2] - C[2] };
        const double lambda = fabs(dot3(xc, DC) / cd / cd);
        if (lambda >= 1) cells[9] += 1; else cells[(unsigned)floor(lambda * 10)] += 1;
    }
    double sum = 0;
    for (int i = 0; i < 10; i++) if (cells[i] > 0) sum = sum + 1;
    return sum / 10 > 0.7;
}

/* LINEextractor::computeLine3d_svd (LineExtractor.cpp:162-184): mean and principal direction of the listed points */
static void compute_line3d(const rpt_t *pts, const int *idx, int n, double mean[3], double drct[3])
{
    mean[0] = mean[1] = mean[2] = 0;
    for (int i = 0; i < n; i++) { mean[0] = mean[0] + pts[idx[i]].pos[0]; mean[1] = mean[1] + pts[idx[i]].pos[1]; mean[2] = mean[2] + pts[idx[i]].pos[2]; }
    const double s = 1.0 / n;
    mean[0] = mean[0] * s; mean[1] = mean[1] * s; mean[2] = mean[2] * s;
    double S[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    for (int i = 0; i < n; i++) {
        const double q[3] = { pts[idx[i]].pos[0] - mean[0], pts[idx[i]].pos[1] - mean[1], pts[idx[i]].pos[2] - mean[2] };
        for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) S[a][b] += q[a] * q[b];
    }
    S[1][0] = S[0][1]; S[2][0] = S[0][2]; S[2][1] = S[1][2];
    double w[3], V[3][3];
    orc_eig33sym(S, w, V);
    drct[0] = V[0][2]; drct[1] = V[1][2]; drct[2] = V[2][2];      /* right singular vector of the largest singular value */
}

/* Frame::isLineGood for n key lines.  depth: raw u16 with depth_factor (imDepth = u16 * factor as float, Tracking.cc:156-160). */
int orc_lines_3d(const orc_keyline *kl, int n, const uint16_t *depth, int w, int h, int stride_bytes,
                 float fx, float fy, float cx, float cy, float depth_factor, uint32_t seed, orc_line3d *out)
{
    (void)fy;
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;
    int good = 0;
    for (int li = 0; li < n; li++) {
        orc_line3d *o = &out[li];
        memset(o, 0, sizeof(*o));
        o->line_eq[0] = o->line_eq[1] = o->line_eq[2] = -1.0f;
        o->line_nor[0] = o->line_nor[1] = o->line_nor[2] = -1.0;
        const float sx = kl[li].sx, sy = kl[li].sy, ex = kl[li].ex, ey = kl[li].ey;
        const float dxf = sx - ex, dyf = sy - ey;
        const double len = sqrt((double)dxf * dxf + (double)dyf * dyf);
        int nsmp = (int)len < 20 ? (int)len : 20;
        const double numSmp = (double)nsmp;
        rpt_t pts[L3_MAXP]; int np = 0;
        if (nsmp >= 1) for (int j = 0; j <= nsmp; j++) {
            const double a = 1 - j / numSmp, b = j / numSmp;
            const float px = (float)((double)sx * a) + (float)((double)ex * b), py = (float)((double)sy * a) + (float)((double)ey * b);
            const double ptx = px, pty = py;
            if (ptx < 0 || pty < 0 || ptx >= w || pty >= h) continue;
            int row, col;
            if (floor(ptx) == ptx && floor(pty) == pty) { col = (int)(ptx - 1) > 0 ? (int)(ptx - 1) : 0; row = (int)(pty - 1) > 0 ? (int)(pty - 1) : 0; }
            else { col = (int)ptx; row = (int)pty; }
            if (row < 0 || col < 0 || row >= w || col >= h) continue;             /* sic: row against cols, col against rows */
            const uint16_t raw = *(const uint16_t *)((const uint8_t *)depth + (size_t)row * stride_bytes + 2 * (size_t)col);
            const float df = (float)raw * depth_factor;
            if (df <= 0.01) continue;
            double p[3];
            p[2] = df;
            p[0] = ((float)col - cx) * p[2] * invfx;
            p[1] = ((float)row - cy) * p[2] * invfy;
            comp_pt3d_cov(p, (double)fx, &pts[np]); np++;
        }
        o->n_samples = np;
        if (np < 5) continue;
        /* extract3dline_mahdist */
        const int pairs = (int)(np * (np - 1) * 0.5);
        const int maxIter = pairs < 10 ? pairs : 10;
        const double distThresh = 3.0;
        int indexes[L3_MAXP]; for (int i = 0; i < np; i++) indexes[i] = i;
        int best[L3_MAXP], nbest = 0, bestA = 0, bestB = 0;
        unsigned rs = seed ^ (0x9E3779B9u * (unsigned)(li + 1)); if (rs == 0) rs = 0x6D2B79F5u;
        for (int it = 0; it < maxIter; it++) {
            int left = np;
            for (int k = 0; k < 2; k++) {                           /* random_unique(begin, end, 2) */
                const int r = k + (int)((xs32(&rs) & 0x7FFFFFFFu) % (unsigned)left);
                const int t = indexes[k]; indexes[k] = indexes[r]; indexes[r] = t;
                left--;
            }
            const rpt_t *A = &pts[indexes[0]], *B = &pts[indexes[1]];
            const double ab[3] = { B->pos[0] - A->pos[0], B->pos[1] - A->pos[1], B->pos[2] - A->pos[2] };
            if (sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]) < 0.0000000001) continue;
            int inl[L3_MAXP], ninl = 0;
            for (int i = 0; i < np; i++) if (mah_dist(&pts[i], A->pos, B->pos) < distThresh) inl[ninl++] = i;
            if (ninl > nbest && verify_3d_line(pts, inl, ninl, A->pos, B->pos)) {
                nbest = ninl; memcpy(best, inl, sizeof(int) * ninl); bestA = indexes[0]; bestB = indexes[1];
            }
            if (nbest > np * 0.6) break;
        }
        double Aout[3] = { 0, 0, 0 }, Bout[3] = { 0, 0, 0 };
        if (nbest >= 2) {
            double m[3], d[3];
            for (int c = 0; c < 3; c++) { m[c] = (pts[bestA].pos[c] + pts[bestB].pos[c]) * 0.5; d[c] = pts[bestB].pos[c] - pts[bestA].pos[c]; }
            for (;;) {
                double tm[3], td[3], q2[3];
                compute_line3d(pts, best, nbest, tm, td);
                q2[0] = tm[0] + td[0]; q2[1] = tm[1] + td[1]; q2[2] = tm[2] + td[2];
                int tmp[L3_MAXP], nt = 0;
                for (int i = 0; i < np; i++) if (mah_dist(&pts[i], tm, q2) < distThresh) tmp[nt++] = i;
                if (nt > nbest) { nbest = nt; memcpy(best, tmp, sizeof(int) * nt); memcpy(m, tm, sizeof(m)); memcpy(d, td, sizeof(d)); }
                else break;
            }
            double minv = 100, maxv = -100; int e1 = 0, e2 = 0;
            for (int i = 0; i < nbest; i++) {
                const double q[3] = { pts[best[i]].pos[0] - m[0], pts[best[i]].pos[1] - m[1], pts[best[i]].pos[2] - m[2] };
                const double dp = dot3(q, d);
                if (dp < minv) { minv = dp; e1 = i; }
                if (dp > maxv) { maxv = dp; e2 = i; }
            }
            memcpy(Aout, pts[best[e1]].pos, sizeof(Aout)); memcpy(Bout, pts[best[e2]].pos, sizeof(Bout));
        }
        o->n_inliers = nbest;
        for (int i = 0; i < nbest; i++) o->inlier_mask |= 1u << best[i];
        const double ab[3] = { Aout[0] - Bout[0], Aout[1] - Bout[1], Aout[2] - Bout[2] };
        if (sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]) > 0.02) {
            memcpy(o->A, Aout, sizeof(Aout)); memcpy(o->B, Bout, sizeof(Bout));
            const float le[3] = { (float)(Bout[0] - Aout[0]), (float)(Bout[1] - Aout[1]), (float)(Bout[2] - Aout[2]) };
            const float magn = sqrtf(le[0] * le[0] + le[1] * le[1] + le[2] * le[2]);
            o->line_eq[0] = le[0] / magn; o->line_eq[1] = le[1] / magn; o->line_eq[2] = le[2] / magn;
            o->line_nor[0] = Aout[1] * Bout[2] - Aout[2] * Bout[1];
            o->line_nor[1] = Aout[2] * Bout[0] - Aout[0] * Bout[2];
            o->line_nor[2] = Aout[0] * Bout[1] - Aout[1] * Bout[0];
            o->good = 1; good++;
        }
    }
    return good;
}

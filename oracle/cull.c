/*
 * cull.c -- oracle for Frame::cullingLine (reference src/Frame.cc:952-1116) and its helpers
 * PointLineDistance (1117-1126), TwoLineAngle (1127-1140), MergeTwoLines (1141-1202): merge near-collinear
 * LSD segments, rebuild the KeyLines, re-sort by response, second LBD pass, line functions.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Determinism rules, same as the rest of the oracle:
 *   - atan / atan2 called on float arguments are evaluated in double on the promoted values;
 *   - std::sort by response is made stable (ties keep the build order);
 *   - cv::LineIterator / cv::clipLine restated from OpenCV 3.2.0 (ASSUMED).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_lbd_compute(const uint8_t *gray, int w, int h, int stride, const orc_keyline *kl, int n, uint8_t *desc32, float *desc72);

static double point_line_distance(const double l[4], float px, float py)                 /* Frame.cc:1117-1126 */
{
    const double x0 = (double)px, y0 = (double)py, x1 = l[0], y1 = l[1], x2 = l[2], y2 = l[3];
    return fabs((y2 - y1) * x0 + (x1 - x2) * y0 + ((x2 * y1) - (x1 * y2))) / sqrt((y2 - y1) * (y2 - y1) + (x1 - x2) * (x1 - x2));
}

static double two_line_angle(const double *f1, const double *f2)                          /* Frame.cc:1127-1140, as written */
{
    double v1[3] = { f1[0], f1[1], f1[2] }, v2[3] = { f2[0], f2[1], f2[2] };
    v1[0] /= v1[2]; v1[1] /= v1[2];
    v2[0] /= v2[2]; v2[1] /= v2[2];
    const double a0 = v1[0] / v1[2], a1 = v1[1] / v1[2], b0 = v2[0] / v2[2], b1 = v2[1] / v2[2];   /* divided by z a second time */
    const double a = a0 * b0 + a1 * b1;
    const double b = sqrt(a0 * a0 + a1 * a1), c = sqrt(b0 * b0 + b1 * b1);
    return fabs(a / (b * c));
}

static void merge_two_lines(const float l1[4], const float l2[4], float out[4])            /* Frame.cc:1141-1202 */
{
    const double PI = 3.1415926535897932384626433832795;                                   /* CV_PI */
    const float ax = l1[0], ay = l1[1], bx = l1[2], by = l1[3], cx = l2[0], cy = l2[1], dx = l2[2], dy = l2[3];
    const float dlix = bx - ax, dliy = by - ay, dljx = dx - cx, dljy = dy - cy;
    const double li = sqrt((double)(dlix * dlix) + (double)(dliy * dliy));
    const double lj = sqrt((double)(dljx * dljx) + (double)(dljy * dljy));
    const double xg = (li * (double)(ax + bx) + lj * (double)(cx + dx)) / (double)(2.0 * (li + lj));
    const double yg = (li * (double)(ay + by) + lj * (double)(cy + dy)) / (double)(2.0 * (li + lj));
    const double thi = dlix == 0.0f ? PI / 2.0 : atan((double)(dliy / dlix));
    const double thj = dljx == 0.0f ? PI / 2.0 : atan((double)(dljy / dljx));
    double thr;
    if (fabs(thi - thj) <= PI / 2.0) thr = (li * thi + lj * thj) / (li + lj);
    else { const double tmp = thj - PI * (thj / fabs(thj)); thr = li * thi + lj * tmp; thr /= (li + lj); }
    const double s = sin(thr), c = cos(thr);
    const double axg = ((double)ay - yg) * s + ((double)ax - xg) * c, bxg = ((double)by - yg) * s + ((double)bx - xg) * c;
    const double cxg = ((double)cy - yg) * s + ((double)cx - xg) * c, dxg = ((double)dy - yg) * s + ((double)dx - xg) * c;
    const double d1 = fmin(axg, fmin(bxg, fmin(cxg, dxg))), d2 = fmax(axg, fmax(bxg, fmax(cxg, dxg)));
    out[0] = (float)(d1 * c + xg); out[1] = (float)(d1 * s + yg); out[2] = (float)(d2 * c + xg); out[3] = (float)(d2 * s + yg);
}

/* cv::clipLine(Size, Point&, Point&), OpenCV 3.2.0 imgproc/src/drawing.cpp (ASSUMED) */
static int clip_line(int w, int h, long long *px1, long long *py1, long long *px2, long long *py2)
{
    long long x1 = *px1, y1 = *py1, x2 = *px2, y2 = *py2;
    const long long right = w - 1, bottom = h - 1;
    if (w <= 0 || h <= 0) return 0;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (a - y1) * (x2 - x1) / (y2 - y1); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
        if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (a - y2) * (x2 - x1) / (y2 - y1); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) { a = c1 == 1 ? 0 : right; y1 += (a - x1) * (y2 - y1) / (x2 - x1); x1 = a; c1 = 0; }
            if (c2) { a = c2 == 1 ? 0 : right; y2 += (a - x2) * (y2 - y1) / (x2 - x1); x2 = a; c2 = 0; }
        }
    }
    *px1 = x1; *py1 = y1; *px2 = x2; *py2 = y2;
    return (c1 | c2) == 0;
}

/* cv::LineIterator(img, Point(p1), Point(p2), 8).count with clipping (Frame.cc:1081-1082) */
int orc_line_iterator_count_clipped(int w, int h, float fx1, float fy1, float fx2, float fy2)
{
    long long x1 = orc_cvround_f(fx1), y1 = orc_cvround_f(fy1), x2 = orc_cvround_f(fx2), y2 = orc_cvround_f(fy2);
    if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
        (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h)
        if (!clip_line(w, h, &x1, &y1, &x2, &y2)) return 0;
    const long long dx = llabs(x2 - x1), dy = llabs(y2 - y1);
    return (int)((dx > dy ? dx : dy) + 1);
}

/* Frame::cullingLine(imGray, dis, angle, endpoint_dis, min_len_pow): kl/fn in (n lines) -> kl/desc/fn out.
 * kl_out needs room for n lines.  Returns the number of lines after merging. */
int orc_cull_lines(const uint8_t *gray, int w, int h, int stride, const orc_keyline *kl, const double *fn, int n,
                   double dis, double angle_deg, double endpoint_dis, orc_keyline *kl_out, uint8_t *desc32, double *fn_out)
{
    if (n <= 0) return 0;
    char *tag = (char *)calloc(n, 1);
    int *grp_start = (int *)calloc(n + 1, sizeof(int)), *grp = (int *)malloc(sizeof(int) * n), ngrp = 0;   /* robust_Line as CSR */
    const double cos_th = cos(angle_deg * 0.0174533);
    for (int i = 0; i < n; i++) {
        grp_start[i] = ngrp;
        if (tag[i]) continue;
        const double v1[4] = { kl[i].sx, kl[i].sy, kl[i].ex, kl[i].ey };
        for (int j = i + 1; j < n; j++) {
            if (tag[j]) continue;
            const double v2[4] = { kl[j].sx, kl[j].sy, kl[j].ex, kl[j].ey };
            /* cv::Point2f arithmetic: sums in float, the product with the double 0.5 rounded back to float */
            const float m12x = (float)((double)(kl[i].sx + kl[i].ex) * 0.5), m12y = (float)((double)(kl[i].sy + kl[i].ey) * 0.5);
            float m21x = (float)((double)(kl[j].ex + kl[j].sx) * 0.5), m21y = (float)((double)(kl[j].ey + kl[j].sy) * 0.5);
            m21x += kl[j].sx; m21y += kl[j].sy;                                            /* Frame.cc:977, as written */
            const double dis12 = point_line_distance(v2, m12x, m12y), dis21 = point_line_distance(v1, m21x, m21y);
            if (!(dis12 < dis || dis21 < dis)) continue;
            if (!(two_line_angle(fn + 3 * i, fn + 3 * j) > cos_th)) continue;
            double bx[4] = { v1[0], v1[2], v2[0], v2[2] }, by[4] = { v1[1], v1[3], v2[1], v2[3] };
            for (int a = 1; a < 4; a++) { double t = bx[a]; int b = a - 1; while (b >= 0 && bx[b] > t) { bx[b + 1] = bx[b]; b--; } bx[b + 1] = t; }
            for (int a = 1; a < 4; a++) { double t = by[a]; int b = a - 1; while (b >= 0 && by[b] > t) { by[b + 1] = by[b]; b--; } by[b + 1] = t; }
            const double dx = bx[3] - bx[0], dy = by[3] - by[0];
            const double dx1 = fabs(v1[0] - v1[2]), dx2 = fabs(v2[0] - v2[2]), dy1 = fabs(v1[1] - v1[3]), dy2 = fabs(v2[1] - v2[3]);
            if (dx > dx1 + dx2 && bx[2] - bx[1] > endpoint_dis) continue;
            if (dy > dy1 + dy2 && by[2] - by[1] > endpoint_dis) continue;
            grp[ngrp++] = j; tag[i] = 1; tag[j] = 1;
        }
    }
    grp_start[n] = ngrp;
    /* merged / surviving segments, in the order of their first line */
    memset(tag, 0, n);
    float (*nl)[4] = (float (*)[4])malloc(sizeof(float) * 4 * n);
    int m = 0;
    for (int i = 0; i < n; i++) {
        float cur[4] = { kl[i].sx, kl[i].sy, kl[i].ex, kl[i].ey };
        const int cnt = grp_start[i + 1] - grp_start[i];
        for (int q = grp_start[i]; q < grp_start[i + 1]; q++) {
            const int j = grp[q];
            const float y1[4] = { kl[j].sx, kl[j].sy, kl[j].ex, kl[j].ey };
            float r[4]; merge_two_lines(cur, y1, r); memcpy(cur, r, sizeof(r));
            tag[j] = 1; tag[i] = 1;
        }
        if (cnt != 0 || !tag[i]) { memcpy(nl[m], cur, sizeof(cur)); m++; }
    }
    /* new KeyLines (Frame.cc:1062-1086) */
    orc_keyline *all = (orc_keyline *)malloc(sizeof(orc_keyline) * (m + 1));
    for (int i = 0; i < m; i++) {
        orc_keyline k; memset(&k, 0, sizeof(k));
        k.sx = k.sox = nl[i][0]; k.sy = k.soy = nl[i][1]; k.ex = k.eox = nl[i][2]; k.ey = k.eoy = nl[i][3];
        const double ddx = (double)(nl[i][0] - nl[i][2]), ddy = (double)(nl[i][1] - nl[i][3]);
        k.length = (float)sqrt(ddx * ddx + ddy * ddy);
        k.octave = 0;
        k.angle = (float)atan2((double)(k.ey - k.sy), (double)(k.ex - k.sx));
        k.size = (k.ex - k.sx) * (k.ey - k.sy);
        k.pt_x = (k.ex + k.sx) / 2; k.pt_y = (k.ey + k.sy) / 2;
        k.num_pixels = orc_line_iterator_count_clipped(w, h, nl[i][0], nl[i][1], nl[i][2], nl[i][3]);
        k.response = k.length / (float)(w > h ? w : h);
        k.class_id = -1;
        all[i] = k;
    }
    for (int i = 1; i < m; i++) {                                                           /* sort by response desc, stable */
        orc_keyline v = all[i]; int j = i - 1;
        while (j >= 0 && all[j].response < v.response) { all[j + 1] = all[j]; j--; }
        all[j + 1] = v;
    }
    for (int i = 0; i < m; i++) all[i].class_id = i;
    memcpy(kl_out, all, sizeof(orc_keyline) * m);
    if (m > 0) orc_lbd_compute(gray, w, h, stride, kl_out, m, desc32, NULL);               /* second LBD pass (Frame.cc:1094-1096) */
    for (int i = 0; i < m; i++) {
        const double sx = kl_out[i].sx, sy = kl_out[i].sy, ex = kl_out[i].ex, ey = kl_out[i].ey;
        const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
        const double nrm = sqrt(l0 * l0 + l1 * l1);
        fn_out[3 * i] = l0 / nrm; fn_out[3 * i + 1] = l1 / nrm; fn_out[3 * i + 2] = l2 / nrm;
    }
    free(tag); free(grp_start); free(grp); free(nl); free(all);
    return m;
}

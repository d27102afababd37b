/*
 * lsd.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * Line-segment detection as the reference's LINEextractor drives it:
 *   LINEextractor::operator()          src/LineExtractor.cpp:329-380
 *   LSDDetector(C)::detectImpl         Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:130-215
 *   checkLineExtremes                  Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:76-102
 *   sort_lines_by_response             include/auxiliar.h:47-52
 *
 * The detector itself is cv::createLineSegmentDetector() of OpenCV 3.2.0 (modules/imgproc/src/lsd.cpp),
 * NOT vendored by the reference.  ASSUMED semantics, restated from the published algorithm
 * (von Gioi et al., LSD) as OpenCV 3.0-3.2 implements it with LSD_REFINE_STD defaults
 * (scale 0.8, sigma_scale 0.6, quant 2, ang_th 22.5, density_th 0.7, 1024 bins, no NFA test):
 *   - the input is converted to CV_64F; GaussianBlur (7x7, sigma 0.75, REFLECT_101) and the 0.8x
 *     INTER_LINEAR resize run in double precision (3.0-3.2 style; later versions keep CV_8U);
 *   - level-line angle = fastAtan2(float(gx), float(-gy)) * pi/180 where the gradient norm exceeds
 *     rho = quant / sin(22.5 deg), else NOTDEF;
 *   - the pseudo-ordered pixel list is built but the seed loop walks the list's storage order,
 *     i.e. raster order (a known quirk of these versions), so no sort enters the result;
 *   - region growing visits the 8-neighbourhood with x as the outer loop;
 *   - cos / sin of float arguments evaluate in double (global ::cos), sqrt follows its argument type.
 * Alternative readings exist in later OpenCV versions: the CV_8U pipeline is a switch here
 * (orc_set_reading(ORC_READING_LSD_8U, 1), oracle.h), the std::sort of seeds is not implemented.  The default (all
 * switches 0) is the parity target of the HIP path.
 *
 * Determinism rule: std::sort by response (unstable) is replaced by a stable sort -- ties keep
 * detection order.
 */
#include "oracle.h"
#define _GNU_SOURCE
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define CV_PI 3.1415926535897932384626433832795
#define M_3_2_PI ((3 * CV_PI) / 2)
#define M_2__PI (2 * CV_PI)
#define NOTDEF (-1024.0)
#define NOTUSED 0
#define USED 1
static const double DEG_TO_RADS = CV_PI / 180;

typedef struct { int x, y; double angle, modgrad; } regpt;
typedef struct { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; } rect_t;

typedef struct {
    int w, h;                  /* scaled image size */
    double *scaled, *angles, *modgrad;
    uint8_t *used;
} lsd_t;

static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}
static int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

/* cv::GaussianBlur on CV_64F, ksize 7: RowFilter<double,double> then SymmColumnFilter<Cast<double,double>> */
static void gaussian_blur_f64(const uint8_t *src, int w, int h, int stride, double *dst, int ksize, double sigma)
{
    double k[33], sum = 0;
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; i++) { double x = i - (ksize - 1) * 0.5; k[i] = exp(scale2x * x * x); sum += k[i]; }
    sum = 1. / sum;
    for (int i = 0; i < ksize; i++) k[i] *= sum;
    const int r = ksize / 2;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            double s0 = k[0] * (double)S[reflect101(x - r, w)];
            for (int i = 1; i < ksize; i++) s0 += k[i] * (double)S[reflect101(x + i - r, w)];
            tmp[(size_t)y * w + x] = s0;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double s0 = k[r] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= r; j++)
                s0 += k[r + j] * (tmp[(size_t)reflect101(y + j, h) * w + x] + tmp[(size_t)reflect101(y - j, h) * w + x]);
            dst[(size_t)y * w + x] = s0;
        }
    free(tmp);
}

/* cv::resize(CV_64F, Size(), fx, fy, INTER_LINEAR): float coefficients, double arithmetic */
static void resize_linear_f64(const double *src, int sw, int sh, double *dst, int dw, int dh, double fxs, double fys)
{
    const double scale_x = 1. / fxs, scale_y = 1. / fys;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    float *alpha = (float *)malloc(sizeof(float) * 2 * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx; alpha[2 * dx] = 1.f - fx; alpha[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        const double b0 = 1.f - fy, b1 = fy;
        const int sy0 = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy, sy1 = sy + 1 < 0 ? 0 : sy + 1 >= sh ? sh - 1 : sy + 1;
        const double *S0 = src + (size_t)sy0 * sw, *S1 = src + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            const double a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            const double t0 = S0[sx] * a0 + S0[sx1] * a1;
            const double t1 = S1[sx] * a0 + S1[sx1] * a1;
            dst[(size_t)dy * dw + dx] = t0 * b0 + t1 * b1;
        }
    }
    free(xofs); free(alpha);
}

static double dist2(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
static double distd(double x1, double y1, double x2, double y2) { return sqrt(dist2(x1, y1, x2, y2)); }
static double angle_diff_signed(double a, double b)
{
    double diff = a - b;
    while (diff <= -CV_PI) diff += M_2__PI;
    while (diff > CV_PI) diff -= M_2__PI;
    return diff;
}
static double angle_diff(double a, double b) { return fabs(angle_diff_signed(a, b)); }

static int is_aligned(const lsd_t *L, int addr, double theta, double prec)
{
    if (addr < 0) return 0;
    const double a = L->angles[addr];
    if (a == NOTDEF) return 0;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) { n_theta -= M_2__PI; if (n_theta < 0) n_theta = -n_theta; }
    return n_theta <= prec;
}

static void region_grow(lsd_t *L, int sx, int sy, regpt *reg, int *reg_size, double *reg_angle, double prec)
{
    *reg_size = 1;
    reg[0].x = sx; reg[0].y = sy;
    int addr = sx + sy * L->w;
    *reg_angle = L->angles[addr];
    reg[0].angle = *reg_angle;
    reg[0].modgrad = L->modgrad[addr];
    float sumdx = (float)cos(*reg_angle);
    float sumdy = (float)sin(*reg_angle);
    L->used[addr] = USED;
    for (int i = 0; i < *reg_size; ++i)
        for (int xx = reg[i].x - 1; xx <= reg[i].x + 1; ++xx)
            for (int yy = reg[i].y - 1; yy <= reg[i].y + 1; ++yy) {
                int c_addr = xx + yy * L->w;
                if ((xx >= 0 && yy >= 0) && (xx < L->w && yy < L->h) && (L->used[c_addr] != USED) &&
                    is_aligned(L, c_addr, *reg_angle, prec)) {
                    L->used[c_addr] = USED;
                    regpt *rp = &reg[*reg_size];
                    rp->x = xx; rp->y = yy;
                    rp->modgrad = L->modgrad[c_addr];
                    const double angle = L->angles[c_addr];
                    rp->angle = angle;
                    ++*reg_size;
                    sumdx += cos((float)angle);       /* float += double */
                    sumdy += sin((float)angle);
                    *reg_angle = orc_fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                }
            }
}

static double get_theta(const regpt *reg, int reg_size, double x, double y, double reg_angle, double prec)
{
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (int i = 0; i < reg_size; ++i) {
        const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
        double dx = regx - x, dy = regy - y;
        Ixx += dy * dy * weight;
        Iyy += dx * dx * weight;
        Ixy -= dx * dy * weight;
    }
    double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)orc_fast_atan2((float)(lambda - Ixx), (float)Ixy)
                                           : (double)orc_fast_atan2((float)Ixy, (float)(lambda - Iyy));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += CV_PI;
    return theta;
}

static void region2rect(const regpt *reg, int reg_size, double reg_angle, double prec, double p, rect_t *rec)
{
    double x = 0, y = 0, sum = 0;
    for (int i = 0; i < reg_size; ++i) {
        const double weight = reg[i].modgrad;
        x += (double)reg[i].x * weight;
        y += (double)reg[i].y * weight;
        sum += weight;
    }
    x /= sum; y /= sum;
    double theta = get_theta(reg, reg_size, x, y, reg_angle, prec);
    double dx = cos(theta), dy = sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = 0; i < reg_size; ++i) {
        double regdx = (double)reg[i].x - x, regdy = (double)reg[i].y - y;
        double l = regdx * dx + regdy * dy;
        double w = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
        if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
    }
    rec->x1 = x + l_min * dx; rec->y1 = y + l_min * dy;
    rec->x2 = x + l_max * dx; rec->y2 = y + l_max * dy;
    rec->width = w_max - w_min;
    rec->x = x; rec->y = y; rec->theta = theta; rec->dx = dx; rec->dy = dy; rec->prec = prec; rec->p = p;
    if (rec->width < 1.0) rec->width = 1.0;
}

static int reduce_region_radius(lsd_t *L, regpt *reg, int *reg_size, double reg_angle, double prec, double p,
                                rect_t *rec, double density, double density_th)
{
    double xc = (double)reg[0].x, yc = (double)reg[0].y;
    double radSq1 = dist2(xc, yc, rec->x1, rec->y1), radSq2 = dist2(xc, yc, rec->x2, rec->y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        for (int i = 0; i < *reg_size; ++i) {
            if (dist2(xc, yc, (double)reg[i].x, (double)reg[i].y) > radSq) {
                L->used[reg[i].x + reg[i].y * L->w] = NOTUSED;
                regpt t = reg[i]; reg[i] = reg[*reg_size - 1]; reg[*reg_size - 1] = t;
                --*reg_size;
                --i;
            }
        }
        if (*reg_size < 2) return 0;
        region2rect(reg, *reg_size, reg_angle, prec, p, rec);
        density = (double)*reg_size / (distd(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    }
    return 1;
}

static int refine(lsd_t *L, regpt *reg, int *reg_size, double reg_angle, double prec, double p, rect_t *rec, double density_th)
{
    double density = (double)*reg_size / (distd(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density >= density_th) return 1;
    double xc = (double)reg[0].x, yc = (double)reg[0].y;
    const double ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (int i = 0; i < *reg_size; ++i) {
        L->used[reg[i].x + reg[i].y * L->w] = NOTUSED;
        if (distd(xc, yc, reg[i].x, reg[i].y) < rec->width) {
            double ang_d = angle_diff_signed(reg[i].angle, ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    double mean_angle = sum / (double)n;
    double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    region_grow(L, reg[0].x, reg[0].y, reg, reg_size, &reg_angle, tau);
    if (*reg_size < 2) return 0;
    region2rect(reg, *reg_size, reg_angle, prec, p, rec);
    density = (double)*reg_size / (distd(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density < density_th) return reduce_region_radius(L, reg, reg_size, reg_angle, prec, p, rec, density, density_th);
    return 1;
}

/* cv::LineSegmentDetector::detect with default parameters; segs = n x 4 floats (x1,y1,x2,y2).
 * Returns the number of segments found (may exceed cap; only cap are written). */
int orc_lsd_detect(const uint8_t *gray, int w, int h, int stride, float *segs, int cap, int *n_out)
{
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, DENSITY_TH = 0.7;
    const double prec = CV_PI * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / sin(prec);
    const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
    const unsigned hk = (unsigned)ceil(sigma * sqrt(2 * sprec * log(10.0)));
    const int ksize = 1 + 2 * hk;
    lsd_t L;
    L.w = orc_cvround_d(w * SCALE); L.h = orc_cvround_d(h * SCALE);
    const size_t np = (size_t)L.w * L.h;
    L.scaled = (double *)malloc(sizeof(double) * np);
    if (orc_get_reading(ORC_READING_LSD_8U)) {
        /* ASSUMED alternative: the detector keeps the image CV_8U -- GaussianBlur and resize(Size(), 0.8, 0.8, INTER_LINEAR)
         * take their u8 fixed-point paths (cvsem.c) and the gradient reads the rounded bytes */
        uint8_t *b8 = (uint8_t *)malloc((size_t)w * h), *s8 = (uint8_t *)malloc(np);
        orc_gaussian_blur_u8(gray, w, h, stride, b8, w, ksize, sigma);
        orc_resize_linear_u8_factor(b8, w, h, w, s8, L.w, L.h, L.w, SCALE, SCALE);
        for (size_t i = 0; i < np; i++) L.scaled[i] = (double)s8[i];
        free(b8); free(s8);
    } else {
        double *blur = (double *)malloc(sizeof(double) * (size_t)w * h);
        gaussian_blur_f64(gray, w, h, stride, blur, ksize, sigma);
        resize_linear_f64(blur, w, h, L.scaled, L.w, L.h, SCALE, SCALE);
        free(blur);
    }
    L.angles = (double *)malloc(sizeof(double) * np);
    L.modgrad = (double *)calloc(np, sizeof(double));
    L.used = (uint8_t *)calloc(np, 1);
    /* ll_angle */
    for (int x = 0; x < L.w; x++) L.angles[(size_t)(L.h - 1) * L.w + x] = NOTDEF;
    for (int y = 0; y < L.h; y++) L.angles[(size_t)y * L.w + L.w - 1] = NOTDEF;
    for (int y = 0; y < L.h - 1; ++y)
        for (int x = 0; x < L.w - 1; ++x) {
            const int addr = y * L.w + x;
            double DA = L.scaled[addr + L.w + 1] - L.scaled[addr];
            double BC = L.scaled[addr + 1] - L.scaled[addr + L.w];
            double gx = DA + BC, gy = DA - BC;
            double norm = sqrt((gx * gx + gy * gy) / 4);
            L.modgrad[addr] = norm;
            if (norm <= rho) L.angles[addr] = NOTDEF;
            else L.angles[addr] = orc_fast_atan2((float)gx, (float)-gy) * DEG_TO_RADS;
        }
    const double LOG_NT = 5 * (log10((double)L.w) + log10((double)L.h)) / 2 + log10(11.0);
    const unsigned min_reg_size = (unsigned)(-LOG_NT / log10(p));
    regpt *reg = (regpt *)malloc(sizeof(regpt) * np);
    int n = 0;
    /* seed loop in list storage (= raster) order over the (w-1) x (h-1) pixels that entered the list */
    for (int y = 0; y < L.h - 1; ++y)
        for (int x = 0; x < L.w - 1; ++x) {
            const int adx = x + y * L.w;
            if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
            int reg_size; double reg_angle;
            region_grow(&L, x, y, reg, &reg_size, &reg_angle, prec);
            if ((unsigned)reg_size < min_reg_size) continue;
            rect_t rec;
            region2rect(reg, reg_size, reg_angle, prec, p, &rec);
            if (!refine(&L, reg, &reg_size, reg_angle, prec, p, &rec, DENSITY_TH)) continue;
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
            if (n < cap) { segs[4 * n] = (float)rec.x1; segs[4 * n + 1] = (float)rec.y1; segs[4 * n + 2] = (float)rec.x2; segs[4 * n + 3] = (float)rec.y2; }
            n++;
        }
    *n_out = n;
    free(reg); free(L.scaled); free(L.angles); free(L.modgrad); free(L.used);
    return 0;
}

/* LSDDetectorC::detectImpl keyline construction for octave 0 (LSDDetector_custom.cpp:161-196) */
static void make_keyline(const float *seg, int w, int h, int class_id, orc_keyline *kl)
{
    float e[4] = { seg[0], seg[1], seg[2], seg[3] };
    /* checkLineExtremes */
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= w) e[0] = (float)w - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= w) e[2] = (float)w - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= h) e[1] = (float)h - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= h) e[3] = (float)h - 1.0f;
    const float octaveScale = 1.0f;                 /* pow((float)scale, 0) */
    kl->sx = e[0] * octaveScale; kl->sy = e[1] * octaveScale; kl->ex = e[2] * octaveScale; kl->ey = e[3] * octaveScale;
    kl->sox = e[0]; kl->soy = e[1]; kl->eox = e[2]; kl->eoy = e[3];
    const double ddx = (double)(e[0] - e[2]), ddy = (double)(e[1] - e[3]);
    kl->length = (float)sqrt(ddx * ddx + ddy * ddy);
    kl->num_pixels = orc_line_iterator_count(w, h, e[0], e[1], e[2], e[3]);
    kl->angle = (float)atan2((double)(kl->ey - kl->sy), (double)(kl->ex - kl->sx));
    kl->class_id = class_id;
    kl->octave = 0;
    kl->size = (kl->ex - kl->sx) * (kl->ey - kl->sy);
    kl->response = kl->length / (float)(w > h ? w : h);
    kl->pt_x = (kl->ex + kl->sx) / 2; kl->pt_y = (kl->ey + kl->sy) / 2;
}

void orc_lbd_compute(const uint8_t *gray, int w, int h, int stride, const orc_keyline *kl, int n,
                     uint8_t *desc32, float *desc72);

/* LINEextractor::operator() (LineExtractor.cpp:329-380): detect, keep the nfeatures strongest,
 * LBD descriptors, normalised 2-D line functions.  Returns total keylines kept in *n_out. */
int orc_line_extract(const uint8_t *gray, int w, int h, int stride, int nfeatures,
                     orc_keyline *kls, uint8_t *desc32, double *linefn3, int cap, int *n_out)
{
    *n_out = 0;
    if (!gray || w <= 0 || h <= 0) return 0;
    int capseg = 16384, nseg = 0;
    float *segs = (float *)malloc(sizeof(float) * 4 * capseg);
    orc_lsd_detect(gray, w, h, stride, segs, capseg, &nseg);
    if (nseg > capseg) nseg = capseg;
    orc_keyline *all = (orc_keyline *)malloc(sizeof(orc_keyline) * (nseg + 1));
    for (int i = 0; i < nseg; i++) make_keyline(segs + 4 * i, w, h, i, &all[i]);
    int n = nseg;
    if (n > nfeatures) {
        /* sort by response descending; stable (ties keep detection order) */
        for (int i = 1; i < n; i++) {
            orc_keyline v = all[i]; int j = i - 1;
            while (j >= 0 && all[j].response < v.response) { all[j + 1] = all[j]; j--; }
            all[j + 1] = v;
        }
        n = nfeatures;
        for (int i = 0; i < n; i++) all[i].class_id = i;
    }
    if (n > cap) n = cap;
    memcpy(kls, all, sizeof(orc_keyline) * n);
    if (n > 0) orc_lbd_compute(gray, w, h, stride, kls, n, desc32, NULL);
    for (int i = 0; i < n; i++) {
        /* sp x ep with homogeneous 1, normalised by the (x,y) norm */
        const double sx = kls[i].sx, sy = kls[i].sy, ex = kls[i].ex, ey = kls[i].ey;
        double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
        const double nrm = sqrt(l0 * l0 + l1 * l1);
        linefn3[3 * i] = l0 / nrm; linefn3[3 * i + 1] = l1 / nrm; linefn3[3 * i + 2] = l2 / nrm;
    }
    *n_out = n;
    free(segs); free(all);
    return 0;
}

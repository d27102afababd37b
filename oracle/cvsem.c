/*
 * cvsem.c -- ORACLE (test infrastructure only; see oracle.h).
 *
 * Restatement of the OpenCV 3.2.0 routines the reference's hot path calls but does not vendor
 * (SURVEY.md Appendix A).  Pinned version: OpenCV 3.2.0 + opencv_contrib 3.2.0
 * (Thirdparty/DBoW2/build/CMakeFiles/DBoW2.dir/link.txt:1).  Everything here is ASSUMED
 * semantics restated from the published algorithms -- "parity unpinned".
 *
 * Reference call sites:
 *   cv::resize            src/ORBextractor.cc:1118 ; (inside cv::LineSegmentDetector)
 *   cv::copyMakeBorder    src/ORBextractor.cc:1120,1125   (REFLECT_101, realised as index reflection)
 *   cv::FAST              src/ORBextractor.cc:807,812
 *   cv::GaussianBlur      src/ORBextractor.cc:1084 ; binary_descriptor_custom.cpp:358
 *   cv::fastAtan2         src/ORBextractor.cc:101
 *   cvRound               src/ORBextractor.cc:79,113,117-118,440,458,1110
 *   cv::Sobel             binary_descriptor_custom.cpp:395-396
 *   cv::LineIterator      LSDDetector_custom.cpp:187
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* cvRound: SSE cvtss2si / lrint under the default rounding mode = round-half-to-even */
int orc_cvround_f(float v) { return (int)lrintf(v); }
int orc_cvround_d(double v) { return (int)lrint(v); }

static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

/* cv::fastAtan2 (OpenCV 3.2 modules/core/src/mathfuncs_core): degrees in [0,360), float poly */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* cv::resize, CV_8UC1, INTER_LINEAR, non-IPP path (IPP linear resize excludes CV_8U in 3.2):
 * 11-bit fixed-point coefficients, horizontal pass into int32, vertical pass with the
 * ((b0*(T0>>4))>>16 + (b1*(T1>>4))>>16 + 2) >> 2 rounding of VResizeLinear<uchar,...>. */
static void resize_linear_u8_impl(const uint8_t *src, int sw, int sh, int sstride,
                                  uint8_t *dst, int dw, int dh, int dstride,
                                  double inv_scale_x, double inv_scale_y)
{
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *H = (int *)malloc(sizeof(int) * (size_t)dw * sh);   /* horizontal pass of every source row */
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short(orc_cvround_f((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = sat_short(orc_cvround_f(fx * 2048));
    }
    for (int y = 0; y < sh; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        int *D = H + (size_t)y * dw;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int s1 = sx + 1 < sw ? S[sx + 1] : 0; /* alpha1 is 0 whenever sx+1 is outside */
            D[dx] = S[sx] * ialpha[2 * dx] + s1 * ialpha[2 * dx + 1];
        }
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        short b0 = sat_short(orc_cvround_f((1.f - fy) * 2048));
        short b1 = sat_short(orc_cvround_f(fy * 2048));
        int sy0 = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;
        int sy1 = sy + 1 < 0 ? 0 : sy + 1 >= sh ? sh - 1 : sy + 1;
        const int *T0 = H + (size_t)sy0 * dw, *T1 = H + (size_t)sy1 * dw;
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (T0[x] >> 4)) >> 16) + ((b1 * (T1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(H);
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride)
{
    /* resize(src, dst, dsize): inv_scale = dsize/ssize (double), scale = 1/inv_scale */
    resize_linear_u8_impl(src, sw, sh, sstride, dst, dw, dh, dstride,
                          (double)dw / sw, (double)dh / sh);
}

/* resize(src, dst, Size(), fx, fy): used by cv::LineSegmentDetector (scale 0.8) */
void orc_resize_linear_u8_factor(const uint8_t *src, int sw, int sh, int sstride,
                                 uint8_t *dst, int dw, int dh, int dstride, double fx, double fy)
{
    resize_linear_u8_impl(src, sw, sh, sstride, dst, dw, dh, dstride, fx, fy);
}

static int g_reading[ORC_READING_COUNT];
void orc_set_reading(int which, int on) { if (which >= 0 && which < ORC_READING_COUNT) g_reading[which] = on != 0; }
int orc_get_reading(int which) { return which >= 0 && which < ORC_READING_COUNT ? g_reading[which] : 0; }

/* cv::getGaussianKernel(ksize, sigma, CV_32F) followed by the CV_8U fixed-point conversion of
 * createSeparableLinearFilter (kernel.convertTo(CV_32S, 256)).  sigma>0 only. */
int orc_gaussian_kernel_q8(int ksize, double sigma, int *k)
{
    float cf[33];
    double scale2x = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < ksize; i++) {
        double x = i - (ksize - 1) * 0.5;
        cf[i] = (float)exp(scale2x * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    int isum = 0;
    for (int i = 0; i < ksize; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = orc_cvround_f(cf[i] * 256.f);
        isum += k[i];
    }
    return isum;
}

/* cv::GaussianBlur on CV_8UC1 (OpenCV 3.2, non-IPP since border is REFLECT_101): integer row
 * filter into int32, then SymmColumnFilter<FixedPtCastEx<int,uchar>, SymmColumnVec_32s8u>.
 * ASSUMED x86/SSE2 build: the vector part of the column filter (columns [0, w & ~3)) evaluates
 * sum * 2^-16 in float -- exact here -- and converts with round-half-to-even; the scalar tail
 * (last w%4 columns) uses (sum + 32768) >> 16.  Both saturate to [0,255]. */
void orc_gaussian_blur_u8(const uint8_t *src, int w, int h, int sstride,
                          uint8_t *dst, int dstride, int ksize, double sigma)
{
    if (g_reading[ORC_READING_BLUR_FLOAT]) {
        /* ASSUMED alternative (IPP-style): the float kernel of getGaussianKernel(ksize, sigma, CV_32F), rows then columns
         * accumulated in float in tap order, one round-half-to-even and saturation at the end */
        float kf[33];
        {
            double scale2x = -0.5 / (sigma * sigma), sum = 0;
            for (int i = 0; i < ksize; i++) { double x = i - (ksize - 1) * 0.5; kf[i] = (float)exp(scale2x * x * x); sum += kf[i]; }
            sum = 1. / sum;
            for (int i = 0; i < ksize; i++) kf[i] = (float)(kf[i] * sum);
        }
        const int r = ksize / 2;
        float *tf = (float *)malloc(sizeof(float) * (size_t)w * h);
        for (int y = 0; y < h; y++) {
            const uint8_t *S = src + (size_t)y * sstride;
            for (int x = 0; x < w; x++) {
                float s = 0;
                for (int i = 0; i < ksize; i++) s += kf[i] * (float)S[reflect101(x + i - r, w)];
                tf[(size_t)y * w + x] = s;
            }
        }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float s = 0;
                for (int j = 0; j < ksize; j++) s += kf[j] * tf[(size_t)reflect101(y + j - r, h) * w + x];
                dst[(size_t)y * dstride + x] = sat_u8(orc_cvround_f(s));
            }
        free(tf);
        return;
    }
    int k[33];
    orc_gaussian_kernel_q8(ksize, sigma, k);
    int r = ksize / 2;
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    int *xi = (int *)malloc(sizeof(int) * (w + 2 * r));
    for (int x = -r; x < w + r; x++) xi[x + r] = reflect101(x, w);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        int *T = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < ksize; i++) s += k[i] * S[xi[x + i]];
            T[x] = s;
        }
    }
    int wv = w & ~3;
    for (int y = 0; y < h; y++) {
        const int *R[33];
        for (int j = 0; j < ksize; j++) R[j] = tmp + (size_t)reflect101(y + j - r, h) * w;
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = 0; j < ksize; j++) s += k[j] * R[j][x];
            int q;
            if (x < wv) {              /* float path: RNE(s / 65536) */
                q = s >> 16;
                int rem = s & 0xFFFF;
                if (rem > 32768 || (rem == 32768 && (q & 1))) q++;
            } else {
                q = (s + 32768) >> 16;
            }
            D[x] = sat_u8(q);
        }
    }
    free(tmp); free(xi);
}

/* cv::Sobel(src, dst, CV_16S, dx, dy, 3) with BORDER_DEFAULT (REFLECT_101): exact integers.
 * (dx,dy) = (1,0): [-1 0 1] along x, [1 2 1] along y; (0,1): transposed roles. */
void orc_sobel3_u8_s16(const uint8_t *src, int w, int h, int sstride,
                       int16_t *dst, int dstride_elems, int dx, int dy)
{
    static const int kd[3] = { -1, 0, 1 }, ks[3] = { 1, 2, 1 };
    const int *kx = dx ? kd : ks, *ky = dy ? kd : ks;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = 0; j < 3; j++) {
                const uint8_t *S = src + (size_t)reflect101(y + j - 1, h) * sstride;
                int t = 0;
                for (int i = 0; i < 3; i++) t += kx[i] * S[reflect101(x + i - 1, w)];
                s += ky[j] * t;
            }
            dst[(size_t)y * dstride_elems + x] = (int16_t)s;
        }
}

/* ---- FAST-9-16 (OpenCV 3.2 modules/features2d/src/fast.cpp, FAST_t<16>) ---- */
static const int fast_off[16][2] = {
    { 0, 3 }, { 1, 3 }, { 2, 2 }, { 3, 1 }, { 3, 0 }, { 3, -1 }, { 2, -2 }, { 1, -3 },
    { 0, -3 }, { -1, -3 }, { -2, -2 }, { -3, -1 }, { -3, 0 }, { -3, 1 }, { -2, 2 }, { -1, 3 }
};

int orc_fast_score(const uint8_t *p, int stride)
{
    /* cornerScore<16>: d[k] = v - ring[k]; best = max over 16 arcs of 9 of
     * max( min(d in arc), -max(d in arc) ); score = best - 1. */
    int d[25], v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[fast_off[k][0] + fast_off[k][1] * stride];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best = -1000;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { if (d[k + j] < mn) mn = d[k + j]; if (d[k + j] > mx) mx = d[k + j]; }
        if (mn > best) best = mn;
        if (-mx > best) best = -mx;
    }
    return best - 1;
}

int orc_fast9_16(const uint8_t *img, int stride, int vw, int vh, int threshold, int *xys, int cap)
{
    int n = 0;
    if (vw < 7 || vh < 7) return 0;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    /* score image of the view: 0 where not a corner / inside the 3-px frame */
    uint8_t *sc = (uint8_t *)calloc((size_t)vw * vh, 1);
    int off[25];
    for (int k = 0; k < 16; k++) off[k] = fast_off[k][0] + fast_off[k][1] * stride;
    for (int k = 16; k < 25; k++) off[k] = off[k - 16];
    for (int y = 3; y < vh - 3; y++) {
        const uint8_t *row = img + (size_t)y * stride;
        for (int x = 3; x < vw - 3; x++) {
            const uint8_t *p = row + x;
            int v = p[0], lo = v - threshold, hi = v + threshold;
            /* quick rejection exactly as a necessary condition (opposite-pair tests) */
            int a = p[off[0]], b = p[off[8]];
            if (!((a < lo) || (a > hi) || (b < lo) || (b > hi))) continue;
            int is = 0, cnt = 0;
            for (int k = 0; k < 25; k++) { if (p[off[k]] < lo) { if (++cnt > 8) { is = 1; break; } } else cnt = 0; }
            if (!is) {
                cnt = 0;
                for (int k = 0; k < 25; k++) { if (p[off[k]] > hi) { if (++cnt > 8) { is = 1; break; } } else cnt = 0; }
            }
            if (is) sc[(size_t)y * vw + x] = (uint8_t)orc_fast_score(p, stride);
        }
    }
    /* strict 3x3 non-max suppression, raster order */
    for (int y = 3; y < vh - 3; y++)
        for (int x = 3; x < vw - 3; x++) {
            int s = sc[(size_t)y * vw + x];
            if (!s) continue;
            const uint8_t *c = sc + (size_t)y * vw + x;
            if (s > c[-1] && s > c[1] && s > c[-vw - 1] && s > c[-vw] && s > c[-vw + 1] &&
                s > c[vw - 1] && s > c[vw] && s > c[vw + 1]) {
                if (n < cap) { xys[3 * n] = x; xys[3 * n + 1] = y; xys[3 * n + 2] = s; }
                n++;
            }
        }
    free(sc);
    return n < cap ? n : cap;
}

/* cv::LineIterator(img, Point(p1), Point(p2), 8).count; Point2f->Point uses cvRound.  The reference clamps the
 * end points into [0, n-1] as floats first (LSDDetector_custom.cpp:76-102), but an end point in (n-1, n), e.g.
 * 641.5 for n = 642, passes that check and rounds to n: cv::LineIterator then clips the segment (cull.c). */
int orc_line_iterator_count(int w, int h, float x1, float y1, float x2, float y2)
{
    return orc_line_iterator_count_clipped(w, h, x1, y1, x2, y2);
}

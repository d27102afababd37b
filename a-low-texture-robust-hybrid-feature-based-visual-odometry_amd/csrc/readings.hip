// readings.hip -- the alternative readings of two OpenCV calls, in the PRODUCT (round 5; hvo_set_readings, include/hvo.h).
//
// SURVEY.md Appendix A marks two places "(?)" where the author's OpenCV 3.2 build may have computed something else than the default this
// library (and the oracle's default) implements; until round 5 they were switches of the oracle only (oracle.h orc_set_reading):
//   HVO_READING_BLUR_FLOAT   cv::GaussianBlur on CV_8U served by IPP: the FLOAT kernel of getGaussianKernel, rows then columns accumulated
//                            in float in tap order, one round-half-to-even + saturation at the end (instead of the 8-bit fixed-point kernels
//                            [18 34 49 55 ..] / 256 with a rounding per pass).  Reaches ORB's 7x7 sigma-2 blur (src/ORBextractor.cc:1084: the
//                            descriptors change, the key points do not) and LBD's 5x5 sigma-1 blur (binary_descriptor_custom.cpp:358).
//   HVO_READING_LSD_8U       cv::LineSegmentDetector keeping the image CV_8U: its 7x7 sigma-0.75 blur and its 0.8x INTER_LINEAR resize run in
//                            their u8 fixed-point paths and the gradient reads the rounded bytes (instead of the CV_64F pipeline).
// These are compatibility paths, not the measured hot path: plain kernels, a thread per pixel, each call site's default kernel left alone.
// Arithmetic as oracle/cvsem.c states it (orc_gaussian_blur_u8 both ways, resize_linear_u8_impl) and oracle/lsd.c:285-293.
#include "hvo_internal.hpp"
#include <math.h>
#include <string.h>
#include <vector>

static __device__ __forceinline__ int rd_reflect(int p, int n) { if (p < 0) p = -p; if (p >= n) p = 2 * (n - 1) - p; return p < 0 ? 0 : p; }

struct GaussK { int q[7]; float f[7]; int ksize; };

// cv::GaussianBlur(u8, ksize x ksize, sigma, BORDER_REFLECT_101), either reading.  Row sums are formed per output pixel (the value of a row
// sum does not depend on who forms it), then the column pass in tap order.
__global__ __launch_bounds__(256) void k_gblur_u8(const uint8_t *__restrict__ src, size_t sframe, int spitch, int w, int h,
                                                  uint8_t *__restrict__ dst, size_t dframe, int dpitch, GaussK K, int float_reading)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t *S = src + (size_t)blockIdx.z * sframe;
    const int ks = K.ksize, r = ks / 2;
    int xi[7];
    for (int i = 0; i < ks; i++) xi[i] = rd_reflect(x + i - r, w);
    int q;
    if (float_reading) {
        float s = 0;
        for (int j = 0; j < ks; j++) {
            const uint8_t *R = S + (size_t)rd_reflect(y + j - r, h) * spitch;
            float t = 0;
            for (int i = 0; i < ks; i++) t = __fadd_rn(t, __fmul_rn(K.f[i], (float)R[xi[i]]));
            s = __fadd_rn(s, __fmul_rn(K.f[j], t));
        }
        q = __float2int_rn(s);                                  // cvRound: half to even
    } else {
        int s = 0;
        for (int j = 0; j < ks; j++) {
            const uint8_t *R = S + (size_t)rd_reflect(y + j - r, h) * spitch;
            int t = 0;
            for (int i = 0; i < ks; i++) t += K.q[i] * (int)R[xi[i]];
            s += K.q[j] * t;
        }
        if (x < (w & ~3)) { q = s >> 16; const int rem = s & 0xFFFF; if (rem > 32768 || (rem == 32768 && (q & 1))) q++; }      // SymmColumnVec_32s8u: RNE(s / 65536)
        else q = (s + 32768) >> 16;                                                                                              // the scalar tail
    }
    dst[(size_t)blockIdx.z * dframe + (size_t)y * dpitch + x] = (uint8_t)min(max(q, 0), 255);
}

// cv::resize(u8, INTER_LINEAR) from host tables: xofs / (a0 | a1 << 16) per destination column, (sy0 | sy1 << 16) / (b0 | b1 << 16) per row
__global__ __launch_bounds__(256) void k_resize_u8_tab(const uint8_t *__restrict__ src, size_t sframe, int spitch, int sw,
                                                       uint8_t *__restrict__ dst, size_t dframe, int dpitch, int dw, int dh,
                                                       const int *__restrict__ xofs, const int *__restrict__ xalpha, const int *__restrict__ yofs, const int *__restrict__ ybeta)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const uint8_t *S = src + (size_t)blockIdx.z * sframe;
    const int sx = xofs[x], a = xalpha[x], a0 = (short)(a & 0xFFFF), a1 = (short)(a >> 16);
    const int yo = yofs[y], sy0 = yo & 0xFFFF, sy1 = yo >> 16, bb = ybeta[y], b0 = (short)(bb & 0xFFFF), b1 = (short)(bb >> 16);
    const uint8_t *R0 = S + (size_t)sy0 * spitch, *R1 = S + (size_t)sy1 * spitch;
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;                  // (alpha1 is 0 whenever sx + 1 is outside)
    const int t0 = R0[sx] * a0 + (sx + 1 < sw ? R0[sx1] : 0) * a1, t1 = R1[sx] * a0 + (sx + 1 < sw ? R1[sx1] : 0) * a1;
    dst[(size_t)blockIdx.z * dframe + (size_t)y * dpitch + x] = (uint8_t)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
}

// ll_angle on the u8 scaled image (oracle/lsd.c:285-312 with L.scaled = the bytes): records {angle, cos, sin, |grad|} and the defined mask.
// 256 threads = 256 columns of one scaled row = eight mask words.
__global__ __launch_bounds__(256) void k_lsd_grad8(const uint8_t *__restrict__ s8, size_t sframe, int sw, int sh, double4 *__restrict__ px4,
                                                   unsigned *__restrict__ defined, int nwords, double rho)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z, lane = threadIdx.x & 63;
    const uint8_t *S = s8 + (size_t)f * sframe;
    bool def = false;
    if (x < sw - 1 && y < sh - 1) {
        const double v00 = (double)S[(size_t)y * sw + x], v01 = (double)S[(size_t)y * sw + x + 1], v10 = (double)S[(size_t)(y + 1) * sw + x], v11 = (double)S[(size_t)(y + 1) * sw + x + 1];
        const double DA = v11 - v00, BC = v01 - v10;
        const double gx = DA + BC, gy = DA - BC;
        const double m = sqrt((gx * gx + gy * gy) / 4);
        if (!(m <= rho)) {
            const double a = (double)hvo_fatan2_deg((float)gx, (float)-gy) * (3.1415926535897932384626433832795 / 180);
            const double af = (double)(float)a;
            px4[(size_t)f * sh * sw + (size_t)y * sw + x] = make_double4(a, cos(af), sin(af), m);
            def = true;
        }
    }
    const unsigned long long bal = __ballot(def);
    if (x < ((sw + 31) & ~31) && (lane & 31) == 0) defined[(size_t)f * nwords + (size_t)y * ((sw + 31) / 32) + (x >> 5)] = (unsigned)(bal >> (lane & 32));
}

// ---- host ----
// getGaussianKernel(ksize, sigma, CV_32F) and its CV_8U fixed-point conversion, as oracle/cvsem.c orc_gaussian_kernel_q8 / orc_gaussian_blur_u8
static GaussK gauss_kernel(int ksize, double sigma)
{
    GaussK K; memset(&K, 0, sizeof(K)); K.ksize = ksize;
    float cf[7]; double scale2x = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < ksize; i++) { const double x = i - (ksize - 1) * 0.5; cf[i] = (float)exp(scale2x * x * x); sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < ksize; i++) { K.f[i] = (float)(cf[i] * sum); K.q[i] = (int)lrintf(K.f[i] * 256.f); }
    return K;
}

int readings_gblur_enqueue(hipStream_t st, const uint8_t *src, size_t sframe, int spitch, int w, int h, uint8_t *dst, size_t dframe, int dpitch,
                           int nframes, int ksize, double sigma, bool float_reading)
{
    if (ksize > 7 || nframes < 1) return HVO_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_gblur_u8, dim3((w + 255) / 256, h, nframes), dim3(256), 0, st, src, sframe, spitch, w, h, dst, dframe, dpitch, gauss_kernel(ksize, sigma), float_reading ? 1 : 0);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

// tables of cv::resize(src, dst, Size(), f, f, INTER_LINEAR) on CV_8U (oracle/cvsem.c resize_linear_u8_impl), uploaded into d_tab (2 dw + 2 dh ints)
int readings_resize_tables(hipStream_t st, int sw, int sh, int dw, int dh, double factor, int *d_tab)
{
    std::vector<int> t(2 * (size_t)dw + 2 * (size_t)dh);
    const double scale = 1. / factor;
    auto sat_short = [](int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; };
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int a0 = sat_short((int)lrintf((1.f - fx) * 2048)), a1 = sat_short((int)lrintf(fx * 2048));
        t[dx] = sx; t[dw + dx] = (int)((unsigned)(unsigned short)a0 | ((unsigned)(unsigned short)a1 << 16));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = (int)floorf(fy); fy -= sy;
        const int b0 = sat_short((int)lrintf((1.f - fy) * 2048)), b1 = sat_short((int)lrintf(fy * 2048));
        const int sy0 = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy, sy1 = sy + 1 < 0 ? 0 : sy + 1 >= sh ? sh - 1 : sy + 1;
        t[2 * dw + dy] = sy0 | (sy1 << 16); t[2 * dw + dh + dy] = (int)((unsigned)(unsigned short)b0 | ((unsigned)(unsigned short)b1 << 16));
    }
    if (hipMemcpyAsync(d_tab, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return HVO_ERR_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? HVO_OK : HVO_ERR_HIP;      // (t goes out of scope)
}

int readings_resize_enqueue(hipStream_t st, const uint8_t *src, size_t sframe, int spitch, int sw, uint8_t *dst, size_t dframe, int dpitch, int dw, int dh,
                            int nframes, const int *d_tab)
{
    hipLaunchKernelGGL(k_resize_u8_tab, dim3((dw + 255) / 256, dh, nframes), dim3(256), 0, st, src, sframe, spitch, sw, dst, dframe, dpitch, dw, dh,
                       d_tab, d_tab + dw, d_tab + 2 * dw, d_tab + 2 * dw + dh);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

int readings_lsd_grad8_enqueue(hipStream_t st, const uint8_t *s8, size_t sframe, int sw, int sh, double4 *px4, unsigned *defined, int nwords, double rho, int nframes)
{
    hipLaunchKernelGGL(k_lsd_grad8, dim3((((sw + 31) & ~31) + 255) / 256, sh, nframes), dim3(256), 0, st, s8, sframe, sw, sh, px4, defined, nwords, rho);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

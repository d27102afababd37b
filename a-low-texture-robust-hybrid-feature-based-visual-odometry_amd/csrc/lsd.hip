// lsd.hip -- batched LSD line detection + LBD descriptors for gfx950 (MI355X).
//
// Replaces LINEextractor::operator() (reference src/LineExtractor.cpp:329-380):
//   LSDDetector::detect -> cv::LineSegmentDetector (OpenCV 3.2 lsd.cpp, not vendored; semantics as
//                          documented in oracle/lsd.c), KeyLine construction
//                          (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:161-196)
//   keep the nLSDFeature strongest lines (LineExtractor.cpp:351-360)
//   BinaryDescriptor::compute -> LBD (Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:
//                          350-398 blur+Sobel, 1026-1372 computeLBD, 401-412 binary conversion)
//   normalised 2-D line functions (LineExtractor.cpp:367-377)
//
// Kernels
//   k_lsd_blur                    GaussianBlur 7x7 sigma 0.75 on the CV_64F image (row and column pass fused)
//   k_lsd_resize_grad             0.8x INTER_LINEAR resize (double) fused with ll_angle: gradient norm,
//                                 one {angle, cos, sin, |grad|} record per pixel, "defined" bitmask
//   k_lsd_grow                    one wave per frame: raster-order seeds, region_grow, region2rect,
//                                 refine / reduce_region_radius; neighbourhoods of up to 7 pending
//                                 region points are fetched per round (availability mask in global
//                                 memory, one 32-byte record per candidate), the reference's sequential
//                                 add-and-update walk is speculated, verified and committed in batches;
//                                 emits segments, key-lines, the top-N selection and the line functions
//   k_lbd_blur_sobel              GaussianBlur 5x5 sigma 1 (u8 fixed point) fused with Sobel dx, dy (s16);
//                                 k_lbd_blur5 / k_lbd_sobel are the two-kernel formulation (HVO_LBD_SPLIT=1)
//   k_lbd_desc                    one 64-thread workgroup per line: 63 row sums, 9 band sums,
//                                 normalisation, 32-byte binary descriptor
//
// fp64/fp32 steps follow the oracle operation by operation (-ffp-contract=off).
#include "hvo_internal.hpp"
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define LSD_PI 3.1415926535897932384626433832795
#define LSD_NOTDEF (-1024.0)
// (segments per frame before the top-N selection: as many as the scaled image can hold -- pixels / min_reg_size, LsdPlan::maxseg; a fixed
// 4096 turned dense periodic patterns, 4800-6700 segments, into HVO_ERR_CAPACITY where the reference just keeps its 200 longest)
#ifndef LSD_RING
#define LSD_RING 256              // pending region points mirrored in LDS (older ones are read back from reg[])
#endif

// the largest double x with sqrt(x) <= rho, sqrt correctly rounded (host libm and the device's fp64 sqrt both are): ll_angle's
// `norm <= threshold` on the squared norm
static double lsd_sqrt_threshold(double rho)
{
    double c = rho * rho;
    while (sqrt(nextafter(c, INFINITY)) <= rho) c = nextafter(c, INFINITY);
    while (sqrt(c) > rho) c = nextafter(c, 0.0);
    return c;
}

struct LsdPlan {
    int w = 0, h = 0, sw = 0, sh = 0, batch = 0, nfeat = 0, nwords = 0;
    double *d_blur = nullptr;                               // w*h doubles per frame of a chunk
    int chunk = 0;                                          // frames the transient images exist for
    double4 *d_px = nullptr;           // sw*sh x {angle, cos, sin, modgrad}: one 32-byte record per scaled pixel (compact plans: for a chunk only)
    // compact plans (large resident batches): only the pixels that have a gradient angle keep their record, in one pool for the batch
    bool compact = false; double4 *d_pool = nullptr; size_t pool_cap = 0;           // records
    unsigned *d_defmask = nullptr, *d_wprefix = nullptr;                            // per frame: the defined mask (the availability mask is consumed), records before each mask word
    long long *d_fbase = nullptr; int *d_fcount = nullptr; unsigned long long *d_pooltop = nullptr;   // per frame: base in the pool (-1: no room), defined pixels; records asked for so far
    unsigned *d_defined = nullptr;                          // bitmask, nwords per frame
    int *d_reg = nullptr;                                   // sw*sh ints
    float *d_segs = nullptr;                                // maxseg x 4
    hvo_keyline *d_kl_all = nullptr;                        // maxseg
    int maxseg = 0;
    hvo_keyline *d_kl = nullptr; uint8_t *d_desc = nullptr; double *d_fn = nullptr; int *d_nkl = nullptr; int *d_flags = nullptr;
    hvo_keyline *d_kl2 = nullptr; uint8_t *d_desc2 = nullptr; double *d_fn2 = nullptr; int *d_nkl2 = nullptr;   // after cullingLine
    uint8_t *d_b5 = nullptr; short2 *d_dxy = nullptr;       // Sobel (dx, dy) interleaved
    int *d_xofs = nullptr, *d_yofs = nullptr; float *d_xa = nullptr, *d_yb = nullptr;   // resize tables
    double k7[4] = { 0, 0, 0, 0 };                          // gaussian taps (double): k[0..3], symmetric
    int k5[3] = { 0, 0, 0 };
    double gL[21], gG[63];
    float *d_gL = nullptr, *d_gG = nullptr;
    double rho = 0, prec = 0, p = 0; unsigned min_reg = 0;
    double rhoT = 0;                   // the largest x with sqrt(x) <= rho (k_lsd_pre tests the squared magnitude)
    long long *d_stats = nullptr;      // per frame 8 counters (diagnostics: hvo_debug_lsd_stats)
    // k_lsd_grow_async (lsd_async.inc), allocated at its first launch for the first `async_b` frames of the plan
    bool pre_fused = true;             // k_lsd_pre instead of k_lsd_blur + k_lsd_resize_grad (HVO_LSD_PRE_SPLIT=1: the pair, with its fp64 image)
    // tuning variables, read when the plan is built (lsd_build_plan)
    struct { Knob dense, lat, async_w, async_early, async_lds, lat_lds, lbd_split, spin_max; } kn;
    uint8_t *d_b8 = nullptr, *d_s8 = nullptr; int *d_rs8tab = nullptr;      // HVO_READING_LSD_8U: the u8 blurred image and the u8 scaled image of a chunk, the resize tables (readings.hip)
    size_t async_w_cap = 0;            // (frame, worker) pairs the per-worker scratch holds
    int *d_redo = nullptr; int async_last_n = 0, async_last_w = 0;      // frames the one-wave kernel grew again after the async growing gave up; the last async launch
    int async_b = 0; unsigned *d_atags = nullptr; void *d_actl = nullptr; int *d_alists = nullptr, *d_ablk = nullptr, *d_afreg = nullptr; unsigned char *d_ainreg = nullptr; int ainreg_b = 0;
};
static LsdPlan *plan_of(hvo_ctx *ctx) { return (LsdPlan *)ctx->lsd; }

static __device__ __forceinline__ int refl(int p, int n) { if (p < 0) p = -p; if (p >= n) p = 2 * (n - 1) - p; return p < 0 ? 0 : p; }

// cv::fastAtan2, float, evaluated without contraction (same as orb.hip)
static __device__ __forceinline__ float fatan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    // both branches of the reference are the same arithmetic on (smaller, larger): selected, not branched -- lanes of a wave differ
    const float ax = fabsf(x), ay = fabsf(y);
    const bool hi = ax >= ay;
    const float c = __fdiv_rn(hi ? ay : ax, __fadd_rn(hi ? ax : ay, eps)), c2 = __fmul_rn(c, c);
    float r = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    if (!hi) r = __fsub_rn(90.f, r);
    if (x < 0) r = __fsub_rn(180.f, r);
    if (y < 0) r = __fsub_rn(360.f, r);
    return r;
}

// ------------------------------------------------------------------------------------------------
// Gaussian blur 7x7 on the double image.  Row pass: s = k0*S[0]; s += k[i]*S[i] (RowFilter order);
// column pass: s = kc*S[0]; s += k[j]*(S[+j] + S[-j]) (SymmColumnFilter order).
// ------------------------------------------------------------------------------------------------
// Both passes in one kernel: a thread owns a column and LSD_BLUR_ROWS output rows, forms the row-pass value of
// the LSD_BLUR_ROWS + 6 source rows it needs (7 cached byte loads each) and runs the column pass over that
// register window.  The fp64 intermediate image (2.4 MB per frame written and read back) does not exist.
#define LSD_BLUR_ROWS 16
__global__ __launch_bounds__(256) void k_lsd_blur(const uint8_t *__restrict__ gray, size_t gframe, int gpitch,
                                                  double *__restrict__ blur, int w, int h, double k0, double k1, double k2, double k3)
{
    const int x = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    if (x >= w) return;
    const int xm3 = refl(x - 3, w), xm2 = refl(x - 2, w), xm1 = refl(x - 1, w), xp1 = refl(x + 1, w), xp2 = refl(x + 2, w), xp3 = refl(x + 3, w);
    const uint8_t *G = gray + (size_t)f * gframe;
    const int yb = blockIdx.y * LSD_BLUR_ROWS;
    double v[LSD_BLUR_ROWS + 6];                       // row-pass values of rows yb-3 .. yb+ROWS+2 of this column
#pragma unroll
    for (int j = 0; j < LSD_BLUR_ROWS + 6; j++) {
        const uint8_t *S = G + (size_t)refl(min(yb + j - 3, h + 2), h) * gpitch;
        double s = k0 * (double)S[xm3];
        s += k1 * (double)S[xm2];
        s += k2 * (double)S[xm1];
        s += k3 * (double)S[x];
        s += k2 * (double)S[xp1];
        s += k1 * (double)S[xp2];
        s += k0 * (double)S[xp3];
        v[j] = s;
    }
#pragma unroll
    for (int j = 0; j < LSD_BLUR_ROWS; j++) {
        const int y = yb + j;
        if (y >= h) break;
        double s = k3 * v[j + 3];
        s += k2 * (v[j + 4] + v[j + 2]);
        s += k1 * (v[j + 5] + v[j + 1]);
        s += k0 * (v[j + 6] + v[j]);
        blur[((size_t)f * h + y) * w + x] = s;
    }
}

// scaled(y,x) of the 0.8x INTER_LINEAR resize of the blurred double image
static __device__ __forceinline__ double scaled_at(const double *B, int w, int x, int y,
                                                   const int *xofs, const float *xa, const int *yofs, const float *yb, int sw_src)
{
    const int sx = xofs[x], sx1 = min(sx + 1, sw_src - 1);
    const double a0 = (double)xa[2 * x], a1 = (double)xa[2 * x + 1];
    const int yo = yofs[y];
    const int sy0 = yo & 0xFFFF, sy1 = yo >> 16;
    const double b0 = (double)yb[2 * y], b1 = (double)yb[2 * y + 1];
    const double *S0 = B + (size_t)sy0 * w, *S1 = B + (size_t)sy1 * w;
    const double t0 = S0[sx] * a0 + S0[sx1] * a1;
    const double t1 = S1[sx] * a0 + S1[sx1] * a1;
    return t0 * b0 + t1 * b1;
}

// One workgroup = 256 consecutive columns x GRAD_ROWS scaled rows.  A thread walks down its column: it forms the
// scaled value of the next row (one bilinear interpolation of the blurred fp64 image), reads its right neighbour's
// from LDS and has the 2x2 neighbourhood of the current pixel in registers, so a scaled value is interpolated
// (GRAD_ROWS + 1) / GRAD_ROWS times instead of twice.  Only ~15 % of the pixels have a gradient above the threshold:
// they are queued in LDS across rows and the expensive part (double cos / sin of the level-line angle) runs on
// full workgroups of 256 queued pixels (the remainder at the end).
#define GRAD_ROWS 8
__global__ __launch_bounds__(256) void k_lsd_resize_grad(const double *__restrict__ blur, int w, int h, int sw, int sh,
                                                         const int *__restrict__ xofs, const float *__restrict__ xa,
                                                         const int *__restrict__ yofs, const float *__restrict__ yb,
                                                         double4 *__restrict__ px4,
                                                         unsigned *__restrict__ defined, int nwords, double rho)
{
    __shared__ double sv[2][257];
    __shared__ double la[512], lm[512]; __shared__ int lpos[512];
    __shared__ int wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * 256, x = x0 + tid, y0 = blockIdx.y * GRAD_ROWS, f = blockIdx.z;
    const double *B = blur + (size_t)f * w * h;
    double4 *PX = px4 + (size_t)f * sh * sw;
    const bool incol = x < sw, edgecol = x0 + 256 < sw;
    double vprev = incol ? scaled_at(B, w, x, y0, xofs, xa, yofs, yb, w) : 0;
    sv[0][tid] = vprev;
    if (tid == 0) sv[0][256] = edgecol ? scaled_at(B, w, x0 + 256, y0, xofs, xa, yofs, yb, w) : 0;
    __syncthreads();
    double vprev_r = sv[0][tid + 1];
    int npend = 0;
    auto emit = [&](int i) {
        // region_grow accumulates cos(float(angle)) / sin(float(angle)) evaluated in double
        const double aa = la[i], af = (double)(float)aa;
        PX[lpos[i]] = make_double4(aa, cos(af), sin(af), lm[i]);
    };
    for (int r = 0; r < GRAD_ROWS; r++) {
        const int y = y0 + r;
        if (y >= sh) break;
        const bool row1 = y + 1 < sh;
        const int buf = (r + 1) & 1;
        const double vcur = incol && row1 ? scaled_at(B, w, x, y + 1, xofs, xa, yofs, yb, w) : 0;
        sv[buf][tid] = vcur;
        if (tid == 0) sv[buf][256] = edgecol && row1 ? scaled_at(B, w, x0 + 256, y + 1, xofs, xa, yofs, yb, w) : 0;
        __syncthreads();
        const double vcur_r = sv[buf][tid + 1];
        bool def = false;
        double a = LSD_NOTDEF, m = 0;
        if (x < sw - 1 && y < sh - 1) {
            const double DA = vcur_r - vprev, BC = vprev_r - vcur;
            const double gx = DA + BC, gy = DA - BC;
            m = sqrt((gx * gx + gy * gy) / 4);
            if (!(m <= rho)) { a = (double)fatan2_deg((float)gx, (float)-gy) * (LSD_PI / 180); def = true; }
        }
        // records of pixels without a gradient angle are never read (the availability mask filters them): not written
        // 256 threads = 8 words of 32 bits; rows are padded to a multiple of 32 bits in the mask
        const unsigned long long bal = __ballot(def);
        if (x < ((sw + 31) & ~31)) {
            const int word = (y * ((sw + 31) / 32)) + (x >> 5);
            if ((lane & 31) == 0) defined[(size_t)f * nwords + word] = (unsigned)(bal >> (lane & 32));
        }
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int base = 0, total = 0;
        for (int i = 0; i < 4; i++) { const int c = wcnt[i]; if (i < wv) base += c; total += c; }
        if (def) { const int p = npend + base + __popcll(bal & ((1ull << lane) - 1)); la[p] = a; lm[p] = m; lpos[p] = y * sw + x; }
        npend += total;
        __syncthreads();
        if (npend >= 256) {                                   // a full workgroup of queued pixels
            emit(tid);
            const int rem = npend - 256;
            double ta = 0, tm = 0; int tp = 0;
            if (tid < rem) { ta = la[256 + tid]; tm = lm[256 + tid]; tp = lpos[256 + tid]; }
            __syncthreads();
            if (tid < rem) { la[tid] = ta; lm[tid] = tm; lpos[tid] = tp; }
            npend = rem;
            __syncthreads();
        }
        vprev = vcur; vprev_r = vcur_r;
    }
    if (tid < npend) emit(tid);
}

// ------------------------------------------------------------------------------------------------
// k_lsd_pre: the whole preamble in one kernel -- the u8 image in, the 32-byte gradient records and the defined mask out.  The CV_64F
// blurred image (2.46 MB per 640x480 frame written by k_lsd_blur and read back by k_lsd_resize_grad: 5.6 x the algorithmic bytes of
// that pair) is never written.  A workgroup owns a band of PRE_TW scaled columns over PRE_SEG scaled rows and STREAMS down the source
// rows it needs: a thread owns a source column, forms the row-pass value of the next source row (seven cached byte loads, RowFilter's
// order), keeps the last seven in registers, and emits one blurred value per step (SymmColumnFilter's order) into a four-row LDS ring;
// as soon as the two source rows a scaled row interpolates from are in the ring the scaled row is formed (scaled_at's expression) and,
// with the previous scaled row, the gradients of that row (ll_angle as in k_lsd_resize_grad: defined-mask words by ballot, the ~15 % of
// the pixels that have an angle queued for the double cos / sin on full workgroups).  No vertical halo but the seven-row start of a
// segment (7 %), no horizontal one at all: every arithmetic step is the unfused pair's, in its order.
// ------------------------------------------------------------------------------------------------
#define PRE_TW 192                                    // scaled columns of a band: six mask words
#define PRE_SEG 96                                    // scaled rows a workgroup walks
__global__ __launch_bounds__(256) void k_lsd_pre(const uint8_t *__restrict__ gray, size_t gframe, int gpitch, int w, int h, int sw, int sh,
                                                 const int *__restrict__ xofs, const float *__restrict__ xa, const int *__restrict__ yofs, const float *__restrict__ yb,
                                                 double4 *__restrict__ px4, unsigned *__restrict__ defined, int nwords, double rhoT,
                                                 double k0, double k1, double k2, double k3, const int *__restrict__ redo_flags)
{
    // redo_flags: only the frames whose growing gave up (flag 4: k_lsd_grow_async's bounded wait) are formed again -- their availability mask was consumed
    if (redo_flags && !(redo_flags[blockIdx.z] & 4)) return;
    __shared__ double bl[4][256];                     // ring of blurred source rows (row & 3), one value per thread's column
    __shared__ double sc[2][PRE_TW + 1];              // the last two scaled rows
    __shared__ double la[512], lm[512]; __shared__ int lpos[512];
    __shared__ int wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * PRE_TW, y0 = blockIdx.y * PRE_SEG, f = blockIdx.z;
    const uint8_t *G = gray + (size_t)f * gframe;
    double4 *PX = px4 + (size_t)f * sh * sw;
    const int xe = min(x0 + PRE_TW, sw - 1), ye = min(y0 + PRE_SEG, sh - 1);          // last scaled column / row the band reads
    const int nsc = xe - x0 + 1;                                                       // scaled values per row (the gradient's right neighbour included)
    const int cA = xofs[x0], cB = min(xofs[xe] + 1, w - 1), ncols = cB - cA + 1;      // source columns (host-checked: <= 256)
    const int sya = yofs[y0] & 0xFFFF, syb = yofs[ye] >> 16;
    const bool colt = tid < ncols;
    const int x = min(cA + tid, w - 1);
    // unsigned column offsets: a row's seven byte loads take the row pointer as a scalar base and the column as a 32-bit lane offset
    unsigned xm3 = (unsigned)refl(x - 3, w), xm2 = (unsigned)refl(x - 2, w), xm1 = (unsigned)refl(x - 1, w), x00 = (unsigned)x, xp1 = (unsigned)refl(x + 1, w), xp2 = (unsigned)refl(x + 2, w), xp3 = (unsigned)refl(x + 3, w);
    // my scaled column's taps (threads < nsc) -- fixed over the rows
    const int sxq = x0 + min(tid, nsc - 1);
    const int tx0 = xofs[sxq] - cA, tx1 = min(xofs[sxq] + 1, w - 1) - cA;
    const double a0 = (double)xa[2 * sxq], a1 = (double)xa[2 * sxq + 1];
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;
    int ys = y0;                                        // next scaled row to form
    int npend = 0;
    // What a pixel WITH an angle needs beyond the test -- the square root of its squared gradient magnitude, the arctangent, the double
    // cos / sin -- is computed here, on the compacted queue (full workgroups of the ~15 % of the pixels that have one), not on every pixel:
    // the queue holds (float gx, float gy) and the squared magnitude.
    auto emit = [&](int i) {
        const float2 gq = reinterpret_cast<const float2 *>(la)[i];
        const double m = sqrt(lm[i]);
        const double aa = (double)fatan2_deg(gq.x, -gq.y) * (LSD_PI / 180);
        // region_grow accumulates cos(float(angle)) / sin(float(angle)) evaluated in double
        const double af = (double)(float)aa;
        double sn, cs; sincos(af, &sn, &cs);
        PX[lpos[i]] = make_double4(aa, cs, sn, m);
    };
    for (int j = sya - 3; j <= syb + 3; j++) {
        // ---- row pass of source row j (reflected), the window of seven, the column pass of row j - 3 ----
        {
            const uint8_t *S = G + (size_t)refl(min(j, h + 2), h) * gpitch;
            // (the empty asm keeps the zero-extension of a lane offset in this block: the loads then take scalar base + 32-bit lane offset
            // instead of seven 64-bit address additions per row)
            asm("" : "+v"(xm3), "+v"(xm2), "+v"(xm1), "+v"(x00), "+v"(xp1), "+v"(xp2), "+v"(xp3));
            const unsigned q0 = S[xm3], q1 = S[xm2], q2 = S[xm1], q3 = S[x00], q4 = S[xp1], q5 = S[xp2], q6 = S[xp3];
            __builtin_amdgcn_sched_barrier(0);                 // all seven in flight before the first is converted
            double s = k0 * (double)q0;
            s += k1 * (double)q1;
            s += k2 * (double)q2;
            s += k3 * (double)q3;
            s += k2 * (double)q4;
            s += k1 * (double)q5;
            s += k0 * (double)q6;
            v0 = v1; v1 = v2; v2 = v3; v3 = v4; v4 = v5; v5 = v6; v6 = s;
        }
        const int rb = j - 3;                           // the blurred row that is complete now
        if (rb < sya) continue;                         // (uniform) the window is filling
        {
            double s = k3 * v3;
            s += k2 * (v4 + v2);
            s += k1 * (v5 + v1);
            s += k0 * (v6 + v0);
            if (colt) bl[rb & 3][tid] = s;
        }
        __syncthreads();
        // ---- every scaled row whose lower source row is rb (0, 1 or 2 of them) ----
        while (ys <= ye && (yofs[ys] >> 16) <= rb) {     // uniform
            const int yo = yofs[ys];
            const int sy0 = yo & 0xFFFF, sy1 = yo >> 16;
            if (tid < nsc) {
                const double b0 = (double)yb[2 * ys], b1 = (double)yb[2 * ys + 1];
                const double *S0 = bl[sy0 & 3], *S1 = bl[sy1 & 3];
                const double t0 = S0[tx0] * a0 + S0[tx1] * a1;
                const double t1 = S1[tx0] * a0 + S1[tx1] * a1;
                sc[ys & 1][tid] = t0 * b0 + t1 * b1;
            }
            __syncthreads();
            if (ys > y0) {
                // ---- ll_angle of scaled row ys - 1 (its lower neighbours are row ys) ----
                const int y = ys - 1, xg = x0 + tid;
                // ll_angle's test `sqrt(x) <= rho` (x = (gx^2 + gy^2) / 4) is taken on x: rhoT is the largest double whose correctly rounded
                // square root is <= rho (lsd_sqrt_threshold, host), so the two tests agree on every x
                bool def = false;
                double xq = 0; float fgx = 0, fgy = 0;
                if (tid < PRE_TW && xg < sw - 1 && y < sh - 1) {
                    const double vprev = sc[y & 1][tid], vprev_r = sc[y & 1][tid + 1], vcur = sc[ys & 1][tid], vcur_r = sc[ys & 1][tid + 1];
                    const double DA = vcur_r - vprev, BC = vprev_r - vcur;
                    const double gx = DA + BC, gy = DA - BC;
                    xq = (gx * gx + gy * gy) / 4;
                    def = !(xq <= rhoT);
                    fgx = (float)gx; fgy = (float)gy;
                }
                const unsigned long long bal = __ballot(def);
                if (tid < PRE_TW && xg < ((sw + 31) & ~31)) {
                    const int word = (y * ((sw + 31) / 32)) + (xg >> 5);
                    if ((lane & 31) == 0) defined[(size_t)f * nwords + word] = (unsigned)(bal >> (lane & 32));
                }
                // the ~15 % of the pixels that have an angle are queued; the double cos / sin run on full workgroups (all four waves at once:
                // a per-wave queue was tried -- the waves are tied by the barriers of the row loop and wait for whichever one is emitting)
                if (lane == 0) wcnt[wv] = __popcll(bal);
                __syncthreads();
                int base = 0, total = 0;
                for (int i = 0; i < 4; i++) { const int cn = wcnt[i]; if (i < wv) base += cn; total += cn; }
                if (def) { const int p = npend + base + __popcll(bal & ((1ull << lane) - 1)); reinterpret_cast<float2 *>(la)[p] = make_float2(fgx, fgy); lm[p] = xq; lpos[p] = y * sw + xg; }
                npend += total;
                if (npend >= 256) {                       // a full workgroup of queued pixels (npend is the same in every thread)
                    __syncthreads();
                    emit(tid);
                    const int rem = npend - 256;
                    double ta = 0, tm = 0; int tp = 0;
                    if (tid < rem) { ta = la[256 + tid]; tm = lm[256 + tid]; tp = lpos[256 + tid]; }
                    __syncthreads();
                    if (tid < rem) { la[tid] = ta; lm[tid] = tm; lpos[tid] = tp; }
                    npend = rem;
                }
            }
            __syncthreads();                              // the scaled row before this one (and the queue) may be written again
            ys++;
        }
    }
    if (tid < npend) emit(tid);
}

// ------------------------------------------------------------------------------------------------
// Compact records (plans of large resident batches): k_lsd_resize_grad writes a chunk's records into the chunk's dense image as ever;
// k_lsd_prefix keeps the frame's defined mask and the count of records before every mask word, k_lsd_bases gives the chunk's frames their
// places in the batch's pool, k_lsd_compact moves the records there (raster order).  32 bytes per DEFINED pixel (13-20 % of the pixels)
// instead of per pixel: 6.3 -> ~1.7 MB per 640x480 frame, 25 -> ~7 MB at 1280x960.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lsd_prefix(const unsigned *__restrict__ defined, unsigned *__restrict__ defmask, unsigned *__restrict__ wprefix,
                                                    int *__restrict__ fcount, int nwords)
{
    __shared__ int wsum[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned *D = defined + (size_t)f * nwords; unsigned *M = defmask + (size_t)f * nwords, *W = wprefix + (size_t)f * nwords;
    int run = 0;
    for (int base = 0; base < nwords; base += 256) {
        const int i = base + tid;
        const unsigned m = i < nwords ? D[i] : 0u;
        const int c = __popc(m);
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int off = run, tot = 0;
        for (int k = 0; k < 4; k++) { const int v = wsum[k]; if (k < wv) off += v; tot += v; }
        if (i < nwords) { M[i] = m; W[i] = (unsigned)(off + incl - c); }
        run += tot;
        __syncthreads();
    }
    if (tid == 0) fcount[f] = run;
}
__global__ __launch_bounds__(512) void k_lsd_bases(const int *__restrict__ fcount, int m, long long *__restrict__ fbase, unsigned long long *__restrict__ pooltop,
                                                   unsigned long long cap)
{
    __shared__ unsigned long long s[512];
    const int tid = threadIdx.x;
    const unsigned long long c = tid < m ? (unsigned long long)fcount[tid] : 0ull;
    s[tid] = c;
    __syncthreads();
    for (int o = 1; o < 512; o <<= 1) { const unsigned long long t = tid >= o ? s[tid - o] : 0ull; __syncthreads(); s[tid] += t; __syncthreads(); }
    const unsigned long long base0 = pooltop[0], b = base0 + s[tid] - c;
    if (tid < m) fbase[tid] = b + c <= cap ? (long long)b : -1ll;
    __syncthreads();
    if (tid == 511) pooltop[0] = base0 + s[511];              // what the batch asks for so far (lsd_run compares it with the pool's size)
}
__global__ __launch_bounds__(256) void k_lsd_compact(const double4 *__restrict__ dense, size_t nsp, const unsigned *__restrict__ defmask, const unsigned *__restrict__ wprefix,
                                                     const long long *__restrict__ fbase, double4 *__restrict__ pool, int nwords, int sw, int wpr)
{
    const int f = blockIdx.y;
    const long long base = fbase[f];
    if (base < 0) return;
    const double4 *D = dense + (size_t)f * nsp; double4 *P = pool + base;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < nwords; w += gridDim.x * 256) {
        unsigned m = defmask[(size_t)f * nwords + w]; unsigned k = wprefix[(size_t)f * nwords + w];
        const int y = w / wpr, x0 = (w - y * wpr) * 32;
        while (m) { const int b = __ffs((int)m) - 1; m &= m - 1; P[k++] = D[(size_t)y * sw + x0 + b]; }
    }
}

// ------------------------------------------------------------------------------------------------
// k_lsd_grow: the serial heart of LSD, one wave per frame
// ------------------------------------------------------------------------------------------------
struct GrowArgs {
    const double4 *px4; unsigned *avail; int *reg; float *segs; const int *perm;
    const unsigned *defmask, *wprefix; const long long *fbase;      // compact plans: px4 is the pool
    hvo_keyline *kl_all, *kl; double *fn; int *nkl; int *flags; long long *stats;
    int sw, sh, nwords, w, h, nfeat, kl_cap, maxseg;
    double rho, prec, p; unsigned min_reg;
    int redo;                 // 1: only frames whose flags carry 4 (the async growing gave up on them) are grown, the others return at once
    int *redo_count;          // ... and counted (hvo_lsd_async_report)
};

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy; };

struct GrowState {
    const double4 *px4; int *reg; unsigned *avail; int *ring;     // px4: {angle, cos, sin, modgrad}
    const unsigned *defmask, *wprefix;                            // compact plans (rec_index)
    int sw, sh, wpr;          // wpr = mask words per row
    bool lm;                  // the mask lives in LDS (lsd_lds_mask)
#ifdef HVO_LSD_TIMING
    long long t_gather, t_add, n_rounds;      // diagnostics build (tools/lsd_timing.py)
#endif
};

// The "available" mask (bit = pixel has a gradient angle and is not in a region yet) lives in GLOBAL memory:
// k_lsd_resize_grad writes it as the "defined" mask, k_lsd_grow consumes it in place.  Keeping it out of
// LDS (24.5 KB per frame at 640x480) is what lets a CU hold 20+ frames instead of 5 -- the kernel is bound
// by one wave's dependent-instruction latency, so frames in flight are its only source of throughput --
// and leaves the LDS to the kernels that need it (k_fast_cells, k_peac_flood).  Reads bypass the per-CU
// L1 (agent-scope atomic loads); updates are L2 atomics without return, made visible to the wave's later
// reads by the vmcnt(0) of the __syncthreads() that closes every growing round.
// Latency variant (k_lsd_grow_lat, small batches / the streamed mode): the frame's mask is copied into LDS at the start (24.5 KB at
// 640x480, 98 KB at 1280x960) and every test / update of a round is an LDS access instead of an L2 round trip: with one frame per CU
// the LDS is free anyway, and the round's chain shrinks to the one fetch of the candidates' records.  S.lm selects it (uniform).
extern __shared__ unsigned lsd_lds_mask[];
static __device__ __forceinline__ unsigned avail_word(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ bool used_get(const GrowState &S, int x, int y)
{
    const int wi = y * S.wpr + (x >> 5);
    const unsigned wv = S.lm ? lsd_lds_mask[wi] : avail_word(&S.avail[wi]);
    return !((wv >> (x & 31)) & 1u);
}
static __device__ __forceinline__ void used_set(const GrowState &S, int x, int y)
{
    const int wi = y * S.wpr + (x >> 5);
    if (S.lm) atomicAnd(&lsd_lds_mask[wi], ~(1u << (x & 31)));
    else __hip_atomic_fetch_and(&S.avail[wi], ~(1u << (x & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void used_clr(const GrowState &S, int x, int y)
{
    const int wi = y * S.wpr + (x >> 5);
    if (S.lm) atomicOr(&lsd_lds_mask[wi], 1u << (x & 31));
    else __hip_atomic_fetch_or(&S.avail[wi], 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Where pixel (x, y)'s record is.  Compact plans keep records only for pixels with a gradient angle, in raster order: the index is the
// count of defined pixels before it = the prefix of its mask word + the bits below it in the word.  Both words are read beside the
// availability word the caller tests anyway, so the record stays ONE dependent access away (what an index image would not give).
template <bool CP> static __device__ __forceinline__ size_t rec_index(const GrowState &S, int x, int y)
{
    if (!CP) return (size_t)x + (size_t)y * S.sw;
    const int wi = y * S.wpr + (x >> 5);
    return (size_t)S.wprefix[wi] + (size_t)__popc(S.defmask[wi] & ((1u << (x & 31)) - 1u));
}

static __device__ __forceinline__ double angle_diff_signed(double a, double b)
{
    double diff = a - b;
    if (fabs(diff) < 3 * LSD_PI) {                             // the rule (angles in [0, 2 pi)): one correction at most, selected
        diff = diff <= -LSD_PI ? diff + 2 * LSD_PI : diff;
        return diff > LSD_PI ? diff - 2 * LSD_PI : diff;
    }
    while (diff <= -LSD_PI) diff += 2 * LSD_PI;
    while (diff > LSD_PI) diff -= 2 * LSD_PI;
    return diff;
}

// region_grow (OpenCV 3.2 lsd.cpp).  Region points are stored packed (y << 16 | x).
// Per round the wave fetches the 3x3 neighbourhoods of up to 7 pending region points (4 slots of 64
// lanes, slot-major = the reference's visiting order: point index, then x outer / y inner).  The
// sequential decisions are then taken without a scalar scan: every lane holds "valid & aligned with
// the current region angle" for its neighbour, a ballot yields the first such neighbour at or after the
// cursor -- exactly the next pixel the reference would add -- the region angle is updated uniformly,
// and the remaining lanes re-evaluate against the new angle.  Neighbours tested before the cursor are
// never revisited, like the reference's single pass.
#define GROW_SLOTS 1          // measured: 1 -> 21.5k, 2 -> 21.4k, 4 -> 21.1k frames/s (pending points per round average 3; unused slots still cost code)
static __device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
static __device__ __forceinline__ bool lsd_aligned(double a, double theta, double prec)
{
    const double t = fabs(theta - a), t2 = fabs(t - 2 * LSD_PI);          // selected, not branched: the lanes of a wave differ
    return (t > (3 * LSD_PI) / 2 ? t2 : t) <= prec;
}

template <bool CP>
static __device__ void region_grow_wave(GrowState &S, int seed_xy, int &reg_size, double &reg_angle, double prec)
{
    const int lane = threadIdx.x, sw = S.sw, sh = S.sh;
    const int sx0 = seed_xy & 0xFFFF, sy0 = seed_xy >> 16;
    const double a0 = S.px4[rec_index<CP>(S, sx0, sy0)].x;
    double ra = a0;
    // one argument reduction for both (once per seed, 890 seeds per frame) where registers are free: the lone-frame kernel gains 5 %,
    // the batch kernels lose a wave per SIMD to sincos' 16 registers
    double s0_, c0_;
    if (S.lm) sincos(a0, &s0_, &c0_); else { c0_ = cos(a0); s0_ = sin(a0); }
    float sumdx = (float)c0_, sumdy = (float)s0_;
    if (lane == 0) { S.reg[0] = seed_xy; S.ring[0] = seed_xy; used_set(S, sx0, sy0); }
    int rs = 1;
    __syncthreads();
    int i = 0;
    const int k0 = lane / 9, j0 = lane - 9 * k0, jx = j0 / 3 - 1, jy = j0 - 3 * (j0 / 3) - 1;   // lane -> (point, neighbour)
    while (i < rs) {
#ifdef HVO_LSD_TIMING
        const long long tg0 = clock64();
#endif
        const int cnt = min(7 * GROW_SLOTS, rs - i);
        int c[GROW_SLOTS]; double an[GROW_SLOTS], cs[GROW_SLOTS], sn[GROW_SLOTS]; bool valid[GROW_SLOTS];
#pragma unroll
        for (int s = 0; s < GROW_SLOTS; s++) {
            const int k = s * 7 + k0;                // 63 neighbours (7 points) per slot, lane 63 idles
            c[s] = -1; an[s] = LSD_NOTDEF; cs[s] = 0; sn[s] = 0; valid[s] = false;
            if (s * 7 >= cnt) continue;              // slot has no pending point (uniform)
            if (lane < 63 && k < cnt) {
                const int idx = i + k;
                const int pxy = (rs - idx <= LSD_RING) ? S.ring[idx & (LSD_RING - 1)] : __hip_atomic_load(&S.reg[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int xx = (pxy & 0xFFFF) + jx, yy = (pxy >> 16) + jy;
                // one bit test rejects both the pixels without a gradient angle and the ones already taken; only
                // real candidates fetch their record (angle, cos, sin in one 32-byte access)
                const bool inb = (unsigned)xx < (unsigned)sw && (unsigned)yy < (unsigned)sh;
                if (inb & !used_get(S, inb ? xx : 0, inb ? yy : 0)) {
                    const double4 r = S.px4[rec_index<CP>(S, xx, yy)];
                    c[s] = (yy << 16) | xx; an[s] = r.x; cs[s] = r.y; sn[s] = r.z; valid[s] = true;
                }
            }
        }
#ifdef HVO_LSD_TIMING
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const long long tg1 = clock64();
#endif
#pragma unroll
        for (int s = 0; s < GROW_SLOTS; s++) {
            if (s * 7 >= cnt) break;                 // no pending points in this slot (uniform)
            // The reference walks the slot's neighbours in lane order; each one it adds changes the region
            // angle that the later ones are tested against.  Instead of one wave-wide step per added pixel the
            // walk is SPECULATED: assume every neighbour aligned with the current angle gets added, compute
            // the running sums in that order (a short scalar chain), let every such lane evaluate the angle
            // after its own addition in parallel, then let every lane re-take its decision with the angle
            // that would be in force when the cursor reaches it.  Everything before the first lane whose
            // decision differs from the assumption is exactly what the sequential walk does and is committed
            // at once; the walk resumes at that lane.  Mismatches need a candidate within rounding of the
            // tolerance, so a slot is normally done in one pass.
            unsigned long long todo = ~0ull;         // lanes the cursor has not passed yet
            for (;;) {
                const bool al0 = valid[s] && lsd_aligned(an[s], ra, prec);
                const unsigned long long A = __ballot(al0) & todo;
                if (!A) break;
                // running sums over A in lane order; a later lane looking at the same pixel as an assumed
                // addition cannot be added any more and drops out of the assumption
                float psx = sumdx, psy = sumdy, my_sx = 0, my_sy = 0;
                unsigned long long rem = A, Aeff = 0, dropm = 0;
                while (rem) {
                    const int L = __ffsll((long long)rem) - 1;
                    rem &= rem - 1;
                    const int cA = __builtin_amdgcn_readlane(c[s], L);
                    psx = (float)((double)psx + readlane_f64(cs[s], L));
                    psy = (float)((double)psy + readlane_f64(sn[s], L));
                    if (lane == L) { my_sx = psx; my_sy = psy; }
                    Aeff |= 1ull << L;
                    const unsigned long long same = __ballot(c[s] == cA);
                    rem &= ~same; dropm |= same & ~((2ull << L) - 1);
                }
                const double my_ra = (double)fatan2_deg(my_sy, my_sx) * (LSD_PI / 180);     // meaningful on Aeff lanes
                // angle in force when the cursor reaches this lane: the one after the nearest assumed addition below it
                const unsigned long long below = Aeff & ((1ull << lane) - 1);
                const int src = below ? 63 - __clzll((long long)below) : lane;
                const double ra_src = __shfl(my_ra, src);          // unconditional: every source lane must be active
                const double ra_at = below ? ra_src : ra;
                // a lane that dropped out because of an assumed addition below it stays out (that addition is
                // committed whenever the cursor gets this far)
                const bool spec = (Aeff >> lane) & 1ull;
                const bool dropped = (dropm >> lane) & 1ull;
                const bool act = !dropped && valid[s] && lsd_aligned(an[s], ra_at, prec);
                const unsigned long long mis = __ballot(act != spec) & todo;
                const int m = mis ? __ffsll((long long)mis) - 1 : 64;
                const unsigned long long C = m >= 64 ? Aeff : (Aeff & ((1ull << m) - 1));  // committed additions
                if (C) {
                    const int nC = __popcll(C);
                    if ((C >> lane) & 1ull) {
                        const int pos = rs + __popcll(C & ((1ull << lane) - 1));
                        used_set(S, c[s] & 0xFFFF, c[s] >> 16); S.reg[pos] = c[s]; S.ring[pos & (LSD_RING - 1)] = c[s];
                    }
                    const int last = 63 - __clzll((long long)C);
                    sumdx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_sx), last));
                    sumdy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_sy), last));
                    ra = readlane_f64(my_ra, last);
                    rs += nC;
                    // the committed pixels are no candidates any more, in any slot (only a walk that resumes looks at them again)
                    unsigned long long cc = (m < 64 || GROW_SLOTS > 1) ? C : 0ull;
                    while (cc) {
                        const int L = __ffsll((long long)cc) - 1;
                        cc &= cc - 1;
                        const int cA = __builtin_amdgcn_readlane(c[s], L);
#pragma unroll
                        for (int t = 0; t < GROW_SLOTS; t++) if (c[t] == cA) valid[t] = false;
                    }
                }
                if (m >= 64) break;
                todo = ~((1ull << m) - 1);           // resume at the lane whose decision differed
            }
        }
#ifdef HVO_LSD_TIMING
        S.t_gather += tg1 - tg0; S.t_add += clock64() - tg1; S.n_rounds++;
#endif
        i += cnt;
        // One wave: its LDS operations execute in order, so the latency variant (mask and ring in LDS) needs no barrier here -- and must
        // not have one: __syncthreads() also waits for the region-list store just issued to be acknowledged by L2 (~500 cycles per
        // round).  Region points older than the ring are read back from the list with an L1-bypassing load, long after their store
        // (every round's record fetch is waited for, and vmcnt counts stores and loads in order).
        if (S.lm) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    }
    reg_size = rs; reg_angle = ra;
}

// Ordered fp64 accumulation: the sums of region2rect / get_theta / refine must be added in region order; 64 points are staged in LDS
// per step.  Three ordered sums with lanes 0..2 owning one accumulator each: ONE instruction stream (a read and an add per term) serves the
// three chains, a third of what one lane adding all three would issue -- and a lone wave is bound by issue.  n (uniform) =
// staged terms of this step; the order of additions within each sum is unchanged.
static __device__ __forceinline__ void seq_add_lanes(const double *mine, int n, double &acc)
{
    int q = 0;
    for (; q + 8 <= n; q += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc += mine[q + u];
    }
    for (; q < n; q++) acc += mine[q];
}

template <bool CP>
static __device__ void region2rect_wave(const GrowState &S, int reg_size, double reg_angle, double prec, Rect &rec,
                                        double *b0, double *b1, double *b2)
{
    const int lane = threadIdx.x;
    // weighted centroid: x += px*w; y += py*w; sum += w   (region order)
    const double *mine = lane == 0 ? b0 : lane == 1 ? b1 : b2;
    int a_first = 0; double w_first = 0;
    double acc = 0;                                           // lanes 0..2: x, y, sum
    for (int base = 0; base < reg_size; base += 64) {
        const int i = base + lane;
        double t0 = 0, t1 = 0, t2 = 0;
        if (i < reg_size) {
            const int a = S.reg[i]; const int px = a & 0xFFFF, py = a >> 16; const double wgt = S.px4[rec_index<CP>(S, px, py)].w; t0 = (double)px * wgt; t1 = (double)py * wgt; t2 = wgt;
            if (base == 0) { a_first = a; w_first = wgt; }     // the first 64 points stay in registers for the second pass
        }
        b0[lane] = t0; b1[lane] = t1; b2[lane] = t2;
        __syncthreads();
        if (lane < 3) seq_add_lanes(mine, min(64, reg_size - base), acc);
        __syncthreads();
    }
    double x = __shfl(acc, 0), y = __shfl(acc, 1);
    { const double sum = __shfl(acc, 2); x /= sum; y /= sum; }
    // get_theta: Ixx += dy*dy*w; Iyy += dx*dx*w; Ixy -= dx*dy*w
    acc = 0;                                                  // lanes 0..2: Ixx, Iyy, Ixy
    for (int base = 0; base < reg_size; base += 64) {
        const int i = base + lane;
        double t0 = 0, t1 = 0, t2 = 0;
        if (i < reg_size) {
            int a = a_first; double wgt = w_first;
            if (base != 0) { a = S.reg[i]; wgt = S.px4[rec_index<CP>(S, a & 0xFFFF, a >> 16)].w; }
            const int px = a & 0xFFFF, py = a >> 16;
            const double dx = (double)px - x, dy = (double)py - y;
            t0 = dy * dy * wgt; t1 = dx * dx * wgt; t2 = -(dx * dy * wgt);      // a -= b  ==  a += (-b), exactly
        }
        b0[lane] = t0; b1[lane] = t1; b2[lane] = t2;
        __syncthreads();
        if (lane < 3) seq_add_lanes(mine, min(64, reg_size - base), acc);
        __syncthreads();
    }
    const double Ixx = __shfl(acc, 0), Iyy = __shfl(acc, 1), Ixy = __shfl(acc, 2);
    double theta = 0, dx = 0, dy = 0;
    if (lane == 0) {
        const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fatan2_deg((float)(lambda - Ixx), (float)Ixy) : (double)fatan2_deg((float)Ixy, (float)(lambda - Iyy));
        theta *= (LSD_PI / 180);
        if (fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += LSD_PI;
        dx = cos(theta); dy = sin(theta);
    }
    theta = __shfl(theta, 0); dx = __shfl(dx, 0); dy = __shfl(dy, 0);
    // extents.  The reference's "if (l > l_max) .. else if (l < l_min)" chain with l_max, l_min starting
    // at 0 yields l_max = max(0, max l) and l_min = min(0, min l): a value cannot be both above the
    // running max (>= 0) and below the running min (<= 0), so the else never hides an update.
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int i = lane; i < reg_size; i += 64) {
        const int a = S.reg[i];
        const double rdx = (double)(a & 0xFFFF) - x, rdy = (double)(a >> 16) - y;
        const double l = rdx * dx + rdy * dy, wv = -rdx * dy + rdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l); w_max = fmax(w_max, wv); w_min = fmin(w_min, wv);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l_max = fmax(l_max, __shfl_xor(l_max, o)); l_min = fmin(l_min, __shfl_xor(l_min, o));
        w_max = fmax(w_max, __shfl_xor(w_max, o)); w_min = fmin(w_min, __shfl_xor(w_min, o));
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min; rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy;
    if (rec.width < 1.0) rec.width = 1.0;
}

static __device__ __forceinline__ double rect_density(const Rect &r, int reg_size)
{
    const double d = sqrt((r.x2 - r.x1) * (r.x2 - r.x1) + (r.y2 - r.y1) * (r.y2 - r.y1));
    return (double)reg_size / (d * r.width);
}

// reduce_region_radius's removal loop (OpenCV 3.2 lsd.cpp: `if (dist > radSq) { used = NOTUSED; swap(reg[i], reg[reg_size - 1]); --reg_size; --i; }`)
// walks i upwards; an element outside the radius is swapped with the last one and the element swapped in is examined next.  It was one
// lane walking the list -- a dependent load per element, 517 ns apiece: 3.1 of a frame's 22 ms in the batch kernel, and pure waiting for a
// lone frame.  What the walk leaves in [0, K) (K = the elements inside the radius) has a closed form: the inside elements of [0, K) stay
// where they are, and the r-th hole of [0, K) in ascending order receives the r-th inside element of [K, n) counted from the END; the
// elements outside end up in [K, n) (as a set: nothing reads their order).  So: one pass for the keep masks of the 64-element chunks and
// the count of kept elements before each (LDS), then every hole finds its partner by rank -- a binary search over the chunk counts and a
// select within the chunk's mask -- and the two are swapped.  `mask` (64 entries) and `pre` (65) are LDS; n <= 4096 (the caller keeps
// the walk for longer lists).  Returns K.  release(a) is called once per removed element (order does not matter: bit clears / releases).
static __device__ __forceinline__ int lsd_nth_set_bit(unsigned long long m, int t)
{
    unsigned x = (unsigned)m; int pos = 0;
    { const int c = __popc(x); if (t >= c) { t -= c; pos = 32; x = (unsigned)(m >> 32); } }
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        const unsigned low = x & ((1u << w) - 1u); const int pc = __popc(low);
        if (t >= pc) { t -= pc; x >>= w; pos += w; } else x = low;
    }
    return pos;
}
template <class Rel>
static __device__ int reduce_radius_wave(int *reg, int n, double xc, double yc, double radSq, unsigned long long *mask, int *pre, Rel release)
{
    const int lane = threadIdx.x & 63, nc = (n + 63) >> 6;
    int K = 0;
    for (int c = 0; c < nc; c++) {
        const int i = c * 64 + lane;
        bool keep = false;
        if (i < n) {
            const int a = reg[i];
            const double ddx = (double)(a & 0xFFFF) - xc, ddy = (double)(a >> 16) - yc;
            keep = !(ddx * ddx + ddy * ddy > radSq);
            if (!keep) release(a);
        }
        const unsigned long long km = __ballot(keep);
        if (lane == 0) { mask[c] = km; pre[c] = K; }
        K += __popcll(km);
    }
    if (lane == 0) pre[nc] = K;
    __syncthreads();
    if (K == n || K == 0) return K;
    const int cK = K >> 6, bK = K & 63;
    const int keptFront = pre[cK] + __popcll(mask[cK] & ((1ull << bK) - 1ull));       // cK < nc because K < n
    const int H = K - keptFront;
    for (int c = 0; c * 64 < K; c++) {
        const int i = c * 64 + lane;
        const unsigned long long km = mask[c];
        const unsigned long long holes = ~km & (c == cK ? ((1ull << bK) - 1ull) : ~0ull);
        if ((holes >> lane) & 1ull) {
            const int r = (c * 64 - pre[c]) + __popcll(holes & ((1ull << lane) - 1ull));
            const int g = keptFront + (H - 1 - r);                 // my partner: the g-th kept element of the list (it lies in [K, n))
            int lo = cK, hi = nc - 1;                              // the chunk with pre[c2] <= g < pre[c2 + 1]
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[mid] <= g) lo = mid; else hi = mid - 1; }
            const int j = lo * 64 + lsd_nth_set_bit(mask[lo], g - pre[lo]);
            const int ah = reg[i], aj = reg[j];
            reg[i] = aj; reg[j] = ah;
        }
    }
    return K;
}

// refine + reduce_region_radius (LSD_REFINE_STD).  Returns false when the region is rejected.
template <bool CP>
static __device__ bool refine_wave(GrowState &S, int &reg_size, double reg_angle, double prec, Rect &rec, double density_th,
                                   double *b0, double *b1, double *b2, int *n_addr)
{
    const int lane = threadIdx.x;
    double density = rect_density(rec, reg_size);
    if (density >= density_th) return true;
    const int a0 = S.reg[0];
    const double xc = (double)(a0 & 0xFFFF), yc = (double)(a0 >> 16);
    const double ang_c = S.px4[rec_index<CP>(S, a0 & 0xFFFF, a0 >> 16)].x;
    const double *mine = lane == 0 ? b0 : b1;
    double acc = 0; int n = 0;                                // lanes 0, 1: sum, s_sum
    for (int base = 0; base < reg_size; base += 64) {
        const int i = base + lane;
        double t0 = 0, t1 = 0; bool in = false;
        if (i < reg_size) {
            const int a = S.reg[i]; const int px = a & 0xFFFF, py = a >> 16;
            n_addr[lane] = a;
            const double ddx = (double)px - xc, ddy = (double)py - yc;
            if (sqrt(ddx * ddx + ddy * ddy) < rec.width) {
                const double ang_d = angle_diff_signed(S.px4[rec_index<CP>(S, px, py)].x, ang_c);
                t0 = ang_d; t1 = ang_d * ang_d; in = true;
            }
        }
        b0[lane] = t0; b1[lane] = t1;
        n += __popcll(__ballot(in));
        __syncthreads();
        if (i < reg_size) used_clr(S, n_addr[lane] & 0xFFFF, n_addr[lane] >> 16);
        if (lane < 2) seq_add_lanes(mine, min(64, reg_size - base), acc);
        __syncthreads();
    }
    const double sum = __shfl(acc, 0), s_sum = __shfl(acc, 1);
    double tau = 0;
    if (lane == 0) {
        const double mean_angle = sum / (double)n;
        tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    }
    tau = __shfl(tau, 0);
    region_grow_wave<CP>(S, a0, reg_size, reg_angle, tau);
    if (reg_size < 2) return false;
    region2rect_wave<CP>(S, reg_size, reg_angle, prec, rec, b0, b1, b2);
    density = rect_density(rec, reg_size);
    if (density >= density_th) return true;
    // reduce_region_radius
    const double r1 = (rec.x1 - xc) * (rec.x1 - xc) + (rec.y1 - yc) * (rec.y1 - yc);
    const double r2 = (rec.x2 - xc) * (rec.x2 - xc) + (rec.y2 - yc) * (rec.y2 - yc);
    double radSq = r1 > r2 ? r1 : r2;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        // swap-with-last removal is order dependent: lane 0 walks the region
        __syncthreads();
        if (reg_size <= 4096) {
            reg_size = reduce_radius_wave(S.reg, reg_size, xc, yc, radSq, reinterpret_cast<unsigned long long *>(b0), reinterpret_cast<int *>(b1),
                                          [&](int a) { used_clr(S, a & 0xFFFF, a >> 16); });
        } else {
            if (lane == 0) {                                  // a list longer than the chunk table: the walk as written
                int rs = reg_size;
                for (int i = 0; i < rs; ++i) {
                    const int a = S.reg[i];
                    const double ddx = (double)(a & 0xFFFF) - xc, ddy = (double)(a >> 16) - yc;
                    if (ddx * ddx + ddy * ddy > radSq) {
                        used_clr(S, a & 0xFFFF, a >> 16);
                        const int last = S.reg[rs - 1];
                        S.reg[i] = last; S.reg[rs - 1] = a;
                        --rs; --i;
                    }
                }
                reg_size = rs;
            }
            reg_size = __shfl(reg_size, 0);
        }
        __syncthreads();
        if (reg_size < 2) return false;
        region2rect_wave<CP>(S, reg_size, reg_angle, prec, rec, b0, b1, b2);
        density = rect_density(rec, reg_size);
    }
    return true;
}

// cv::LineIterator(...).count for clamped float end points: Point2f -> Point rounds half to even
// checkLineExtremes (LSDDetector_custom.cpp:76-102) leaves end points in (n-1, n) untouched and Point2f -> Point rounds
// them to n, so cv::LineIterator clips here too (LSDDetector_custom.cpp:187)
// cv::LineIterator(img, Point(p1), Point(p2), 8).count with cv::clipLine (OpenCV 3.2.0, ASSUMED)
static __device__ int cull_line_count(int w, int h, float fx1, float fy1, float fx2, float fy2)
{
    long long x1 = __float2int_rn(fx1), y1 = __float2int_rn(fy1), x2 = __float2int_rn(fx2), y2 = __float2int_rn(fy2);
    if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
        (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h) {
        const long long right = w - 1, bottom = h - 1;
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
        if ((c1 | c2) != 0) return 0;
    }
    const long long dx = llabs(x2 - x1), dy = llabs(y2 - y1);
    return (int)((dx > dy ? dx : dy) + 1);
}

// The end of a frame's detection, by ONE wave: KeyLines of the segments (detection order), the nfeat strongest, the line functions.
// Returns the number of key lines kept.
static __device__ __forceinline__ int lsd_emit_keylines(const GrowArgs &g, int f, const float *segs, int nseg, int &flags)
{
    const int lane = threadIdx.x & 63;
    __syncthreads();
    // ---- KeyLines (LSDDetector_custom.cpp:161-196) ----
    hvo_keyline *all = g.kl_all + (size_t)f * g.maxseg;
    const int w = g.w, h = g.h;
    for (int i = lane; i < nseg; i += 64) {
        float e0 = segs[4 * i], e1 = segs[4 * i + 1], e2 = segs[4 * i + 2], e3 = segs[4 * i + 3];
        if (e0 < 0) e0 = 0; if (e0 >= w) e0 = (float)w - 1.0f;
        if (e2 < 0) e2 = 0; if (e2 >= w) e2 = (float)w - 1.0f;
        if (e1 < 0) e1 = 0; if (e1 >= h) e1 = (float)h - 1.0f;
        if (e3 < 0) e3 = 0; if (e3 >= h) e3 = (float)h - 1.0f;
        hvo_keyline kl;
        kl.sx = e0; kl.sy = e1; kl.ex = e2; kl.ey = e3; kl.sox = e0; kl.soy = e1; kl.eox = e2; kl.eoy = e3;
        const double ddx = (double)__fsub_rn(e0, e2), ddy = (double)__fsub_rn(e1, e3);
        kl.length = (float)sqrt(ddx * ddx + ddy * ddy);
        kl.num_pixels = cull_line_count(g.w, g.h, e0, e1, e2, e3);
        kl.angle = (float)atan2((double)__fsub_rn(kl.ey, kl.sy), (double)__fsub_rn(kl.ex, kl.sx));
        kl.class_id = i; kl.octave = 0;
        kl.size = __fmul_rn(__fsub_rn(kl.ex, kl.sx), __fsub_rn(kl.ey, kl.sy));
        kl.response = __fdiv_rn(kl.length, (float)(w > h ? w : h));
        kl.pt_x = __fdiv_rn(__fadd_rn(kl.ex, kl.sx), 2.f); kl.pt_y = __fdiv_rn(__fadd_rn(kl.ey, kl.sy), 2.f);
        all[i] = kl;
    }
    __syncthreads();
    // ---- keep the nfeat strongest (stable by response desc), class_id = rank ----
    hvo_keyline *out = g.kl + (size_t)f * g.kl_cap;
    int n = nseg;
    if (nseg > g.nfeat) {
        n = g.nfeat;
        for (int i = lane; i < nseg; i += 64) {
            const float r = all[i].response;
            int rank = 0;
            for (int q = 0; q < nseg; q++) { const float rq = all[q].response; rank += (rq > r) || (rq == r && q < i); }
            if (rank < n && rank < g.kl_cap) { hvo_keyline kl = all[i]; kl.class_id = rank; out[rank] = kl; }
        }
    } else {
        for (int i = lane; i < nseg; i += 64) if (i < g.kl_cap) out[i] = all[i];
    }
    if (n > g.kl_cap) { n = g.kl_cap; flags |= 2; }
    __syncthreads();
    // ---- 2-D line functions (LineExtractor.cpp:367-377) ----
    double *fn = g.fn + (size_t)f * g.kl_cap * 3;
    for (int i = lane; i < n; i += 64) {
        const double sx = out[i].sx, sy = out[i].sy, ex = out[i].ex, ey = out[i].ey;
        const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
        const double nrm = sqrt(l0 * l0 + l1 * l1);
        fn[3 * i] = l0 / nrm; fn[3 * i + 1] = l1 / nrm; fn[3 * i + 2] = l2 / nrm;
    }
    return n;
}

// The kernel body; two kernels wrap it (below).
template <bool LM, bool CP>
static __device__ __forceinline__ void lsd_grow_body(const GrowArgs &g)
{
    __shared__ double b0[64], b1[64], b2[64];
    __shared__ int n_addr[64];
    __shared__ int ring[LSD_RING];
    const int f = g.perm ? g.perm[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;       // hvo_frame_perm
    if (g.redo) { if (!(g.flags[f] & 4)) return; if (lane == 0) atomicAdd(g.redo_count, 1); }
    const int sw = g.sw, sh = g.sh, wpr = (sw + 31) / 32, nwords = g.nwords;
    const size_t np = (size_t)sw * sh;
    GrowState S;
    S.px4 = g.px4 + f * np; S.defmask = nullptr; S.wprefix = nullptr;
    if (CP) {                                                  // (a frame whose records found no room: lsd_run has grown the pool and come again before this kernel runs)
        const long long fb = g.fbase[f];
        S.px4 = g.px4 + (fb < 0 ? 0 : fb); S.defmask = g.defmask + (size_t)f * nwords; S.wprefix = g.wprefix + (size_t)f * nwords;
    }
    S.reg = g.reg + f * np; S.avail = g.avail + (size_t)f * nwords; S.ring = ring; S.sw = sw; S.sh = sh; S.wpr = wpr; S.lm = LM;
    if (LM) {
        for (int i = lane; i < nwords; i += 64) lsd_lds_mask[i] = S.avail[i];
        __syncthreads();
    }
    float *segs = g.segs + (size_t)f * g.maxseg * 4;
#ifdef HVO_LSD_TIMING
    S.t_gather = S.t_add = S.n_rounds = 0;
#endif
    int nseg = 0, flags = 0;
    long long st_seeds = 0, st_pts = 0, st_tg = 0, st_tr = 0, st_tf = 0, st_big = 0;
    const long long t_begin = wall_clock64();
    // seeds in raster order: words of the available mask, 64 words per step
    for (int wbase = 0; wbase < nwords; wbase += 64) {
        for (;;) {
            const int wi = wbase + lane;
            unsigned m = 0;
            if (wi < nwords) m = LM ? lsd_lds_mask[wi] : avail_word(&S.avail[wi]);
            const unsigned long long nz = __ballot(m != 0);
            if (!nz) break;
            const int wl = __ffsll((long long)nz) - 1;
            const unsigned mw = __shfl(m, wl);
            const int bit = __ffs((int)mw) - 1;
            const int wsel = wbase + wl;
            const int sy = wsel / wpr, sx = (wsel - sy * wpr) * 32 + bit;
            const int seed = (sy << 16) | sx;
            int reg_size; double reg_angle = 0;
            long long t0 = wall_clock64();
            region_grow_wave<CP>(S, seed, reg_size, reg_angle, g.prec);
            long long t1 = wall_clock64(); st_tg += t1 - t0; st_seeds++; st_pts += reg_size;
            if ((unsigned)reg_size < g.min_reg) continue;
            st_big++;
            Rect rec;
            region2rect_wave<CP>(S, reg_size, reg_angle, g.prec, rec, b0, b1, b2);
            long long t2 = wall_clock64(); st_tr += t2 - t1;
            const bool okr = refine_wave<CP>(S, reg_size, reg_angle, g.prec, rec, 0.7, b0, b1, b2, n_addr);
            st_tf += wall_clock64() - t2;
            if (!okr) continue;
            if (nseg < g.maxseg) {
                if (lane == 0) {
                    double x1 = rec.x1 + 0.5, y1 = rec.y1 + 0.5, x2 = rec.x2 + 0.5, y2 = rec.y2 + 0.5;
                    x1 /= 0.8; y1 /= 0.8; x2 /= 0.8; y2 /= 0.8;
                    segs[4 * nseg] = (float)x1; segs[4 * nseg + 1] = (float)y1; segs[4 * nseg + 2] = (float)x2; segs[4 * nseg + 3] = (float)y2;
                }
                nseg++;
            } else flags |= 1;
        }
    }
    const int n = lsd_emit_keylines(g, f, segs, nseg, flags);
    if (lane == 0) {
        g.nkl[f] = n; g.flags[f] = flags;
        long long *st = g.stats + (size_t)f * 8;
        st[0] = st_seeds; st[1] = st_pts; st[2] = st_big; st[3] = st_tg; st[4] = st_tr; st[5] = st_tf; st[6] = wall_clock64() - t_begin; st[7] = nseg;
#ifdef HVO_LSD_STAT_T0
        st[7] = t_begin;                                               // diagnostics build: when the wave started (100 MHz wall clock)
#endif
#ifdef HVO_LSD_TIMING
        st[2] = S.n_rounds; st[4] = S.t_gather; st[5] = S.t_add;      // diagnostics build: rounds, gather ticks, add ticks
#endif
    }
}

// k_lsd_grow: the compiler's own register budget (91 VGPRs, five waves per SIMD): the fastest single frame (12 ms).
// k_lsd_grow_dense: eight waves per SIMD (64 VGPRs, some spills): the kernel is one dependent chain per frame, so frames in
// flight are its only source of throughput once a batch fills the wave slots; measured 69 -> 58 ms per 8192 frames, but
// 12 -> 18 ms for a lone frame -- hence two kernels and a choice by batch size (lsd_run).
__global__ __launch_bounds__(64) void k_lsd_grow(GrowArgs g) { lsd_grow_body<false, false>(g); }
__global__ __launch_bounds__(64) void k_lsd_grow_lat(GrowArgs g) { lsd_grow_body<true, false>(g); }
__global__ __launch_bounds__(64) void k_lsd_grow_c(GrowArgs g) { lsd_grow_body<false, true>(g); }       // compact records
// (round 5, after the radius walk left the kernel: seven waves per SIMD with 72 registers beat eight with 64 and their spills -- 34.2 -> 32.6 ms
// alone, the step 154.1 -> 153.2 ms on one box; six: 31.4 ms alone but 154.3 for the step, five: 38.9 / 154.0)
#ifndef HVO_WPE_GROW
#define HVO_WPE_GROW 7
#endif
__attribute__((amdgpu_waves_per_eu(HVO_WPE_GROW)))
__global__ __launch_bounds__(64) void k_lsd_grow_dense(GrowArgs g) { lsd_grow_body<false, false>(g); }
__attribute__((amdgpu_waves_per_eu(HVO_WPE_GROW)))
__global__ __launch_bounds__(64) void k_lsd_grow_dense_c(GrowArgs g) { lsd_grow_body<false, true>(g); }

#include "lsd_async.inc"

// ------------------------------------------------------------------------------------------------
// LBD: blur 5x5 (u8 fixed point, same rounding rules as the ORB blur), Sobel, descriptor
// ------------------------------------------------------------------------------------------------
#define LBD_BLUR_ROWS 16
#ifdef HVO_WPE_BLUR5
__attribute__((amdgpu_waves_per_eu(HVO_WPE_BLUR5)))
#endif
__global__ __launch_bounds__(256) void k_lbd_blur5(const uint8_t *__restrict__ gray, size_t gframe, int gpitch,
                                                   uint8_t *__restrict__ out, int w, int h, int k0, int k1, int k2)
{
    const int x = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t *G = gray + (size_t)f * gframe;
    const int xm2 = refl(x - 2, w), xm1 = refl(x - 1, w), xp1 = refl(x + 1, w), xp2 = refl(x + 2, w);
    const int yb = blockIdx.y * LBD_BLUR_ROWS;
    // row-pass values of the LBD_BLUR_ROWS + 4 source rows this column needs, each formed once
    int r[LBD_BLUR_ROWS + 4];
#pragma unroll
    for (int j = 0; j < LBD_BLUR_ROWS + 4; j++) {
        const uint8_t *R = G + (size_t)refl(min(yb + j - 2, h + 1), h) * gpitch;
        r[j] = k0 * (R[xm2] + R[xp2]) + k1 * (R[xm1] + R[xp1]) + k2 * R[x];
    }
    const bool vec = x < (w & ~3);                  // OpenCV's SSE2 column path (round half to even) vs its scalar tail
#pragma unroll
    for (int j = 0; j < LBD_BLUR_ROWS; j++) {
        const int y = yb + j;
        if (y >= h) break;
        const int s = k0 * r[j] + k1 * r[j + 1] + k2 * r[j + 2] + k1 * r[j + 3] + k0 * r[j + 4];
        int q;
        if (vec) { q = s >> 16; const int rem = s & 0xFFFF; if (rem > 32768 || (rem == 32768 && (q & 1))) q++; }
        else q = (s + 32768) >> 16;
        out[((size_t)f * h + y) * w + x] = (uint8_t)min(q, 255);
    }
}

__global__ __launch_bounds__(256) void k_lbd_sobel(const uint8_t *__restrict__ b5, short2 *__restrict__ dxy, int w, int h)
{
    const int x = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    if (x >= w) return;
    const uint8_t *B = b5 + (size_t)f * w * h;
    const int xm = refl(x - 1, w), xp = refl(x + 1, w);
    const int yb = blockIdx.y * LBD_BLUR_ROWS;
    // per source row: horizontal difference and horizontal (1 2 1) sum, each formed once
    int hd[LBD_BLUR_ROWS + 2], hs[LBD_BLUR_ROWS + 2];
#pragma unroll
    for (int j = 0; j < LBD_BLUR_ROWS + 2; j++) {
        const uint8_t *R = B + (size_t)refl(min(yb + j - 1, h), h) * w;
        const int a = R[xm], c = R[x], e = R[xp];
        hd[j] = e - a; hs[j] = a + 2 * c + e;
    }
#pragma unroll
    for (int j = 0; j < LBD_BLUR_ROWS; j++) {
        const int y = yb + j;
        if (y >= h) break;
        const int gx = hd[j] + 2 * hd[j + 1] + hd[j + 2];
        const int gy = hs[j + 2] - hs[j];
        dxy[((size_t)f * h + y) * w + x] = make_short2((short)gx, (short)gy);       // interleaved: the descriptor fetches both with one access
    }
}

// Blur and Sobel in one pass: a thread owns 4 adjacent columns and LBD_BLUR_ROWS output rows.  Per source row it
// fetches three aligned dwords (pixels x0-4 .. x0+7; strips on the left / right image border load from clamped
// addresses and permute the reflected bytes into place, EdgeSel in hvo_internal.hpp) and forms the 5-tap row sums of the SIX columns x0-1 .. x0+4 with v_dot4 (the Sobel
// needs the blurred neighbours of its own four); the last five row sums per column give one blurred row, the last
// three blurred rows one Sobel row.  The u8 blurred image is never written.  Reflection of the blurred image at
// the image border (refl(-1) = 1, refl(n) = n-2) is a substitution of the opposite neighbour.
__global__ __launch_bounds__(256) void k_lbd_blur_sobel(const uint8_t *__restrict__ gray, size_t gframe, int gpitch,
                                                        short2 *__restrict__ dxy, int w, int h, int k0, int k1, int k2)
{
    const int nstrip = (w + 3) >> 2, item = blockIdx.x * 256 + threadIdx.x, f = blockIdx.z;
    const int rb = item / nstrip;
    const int x0 = (item - rb * nstrip) * 4, yb = rb * LBD_BLUR_ROWS;
    if (yb >= h) return;
    const uint8_t *G = gray + (size_t)f * gframe;          // rows are 4-byte aligned (pitch % 64 == 0)
    const EdgeSel es = edge_sel(x0, w);
    const int o0 = max(x0 - 4, 0), o2 = min(x0 + 4, gpitch - 4);     // window loads clamped into the row
    const unsigned kA = (unsigned)k0 | ((unsigned)k1 << 8) | ((unsigned)k2 << 16) | ((unsigned)k1 << 24);
    const int wv4 = w & ~3;
    const int v0 = max(yb - 1, 0), v1 = min(yb + LBD_BLUR_ROWS, h - 1);      // blurred rows formed here
    const int ylast = min(yb + LBD_BLUR_ROWS, h) - 1;
    int rs[5][6];                                         // row sums of the last five source rows (oldest first)
    int hd1[4], hs1[4], hd2[4], hs2[4];                    // Sobel row terms of blurred rows v-1 and v-2
#pragma unroll
    for (int j = 0; j < 4; j++) { hd1[j] = hs1[j] = hd2[j] = hs2[j] = 0; }
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
        for (int c = 0; c < 6; c++) rs[q][c] = 0;
    short2 *out = dxy + (size_t)f * h * w;
    const bool vecst = (w & 3) == 0;                       // 16-byte stores need every row start aligned
    for (int sr = v0 - 2; sr <= v1 + 2; sr++) {
        const uint8_t *S = G + (size_t)refl(min(sr, h + 1), h) * gpitch;
        unsigned W0 = *reinterpret_cast<const uint32_t *>(S + o0), W1 = *reinterpret_cast<const uint32_t *>(S + x0),
                 W2 = *reinterpret_cast<const uint32_t *>(S + o2);
        edge_fix(es, W0, W1, W2);                          // REFLECT_101 at the left / right image border (identity elsewhere)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 6; c++) rs[q][c] = rs[q + 1][c];
        // column c = pixel x0-1+c: taps at window bytes c+1 .. c+5 (window byte 0 = pixel x0-4)
        rs[4][0] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W1, W0, 1), kA, (unsigned)k0 * ((W1 >> 8) & 0xFFu), false);
        rs[4][1] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W1, W0, 2), kA, (unsigned)k0 * ((W1 >> 16) & 0xFFu), false);
        rs[4][2] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W1, W0, 3), kA, (unsigned)k0 * (W1 >> 24), false);
        rs[4][3] = (int)__builtin_amdgcn_udot4(W1, kA, (unsigned)k0 * (W2 & 0xFFu), false);
        rs[4][4] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W2, W1, 1), kA, (unsigned)k0 * ((W2 >> 8) & 0xFFu), false);
        rs[4][5] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(W2, W1, 2), kA, (unsigned)k0 * ((W2 >> 16) & 0xFFu), false);
        const int v = sr - 2;                              // blurred row completed by this source row
        if (v < v0) continue;
        int bl[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const int sv = k0 * (rs[0][c] + rs[4][c]) + k1 * (rs[1][c] + rs[3][c]) + k2 * rs[2][c];
            int q;
            if (x0 - 1 + c < wv4) { q = sv >> 16; const int rem = sv & 0xFFFF; if (rem > 32768 || (rem == 32768 && (q & 1))) q++; }   // SSE2 column path: half to even
            else q = (sv + 32768) >> 16;                                                                                                   // scalar tail
            bl[c] = min(q, 255);
        }
        int hd0[4], hs0[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = x0 + j;
            const int a = x == 0 ? bl[j + 2] : bl[j], e = x == w - 1 ? bl[j] : bl[j + 2];
            hd0[j] = e - a; hs0[j] = a + 2 * bl[j + 1] + e;
        }
        // Sobel row y = v-1 from blurred rows v-2 (reflected at the top: row 1), v-1, v; the bottom row y = h-1 uses row h-2 twice
        for (int pass = 0; pass < 2; pass++) {
            const int y = pass == 0 ? v - 1 : v;
            if (pass == 1 && !(v == h - 1)) break;
            if (y < yb || y > ylast) continue;
            short2 g[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int gx, gy;
                if (pass == 0) {
                    const int hdu = v >= 2 ? hd2[j] : hd0[j], hsu = v >= 2 ? hs2[j] : hs0[j];
                    gx = hdu + 2 * hd1[j] + hd0[j]; gy = hs0[j] - hsu;
                } else { gx = 2 * hd1[j] + 2 * hd0[j]; gy = 0; }         // rows (h-2, h-1, h-2)
                g[j] = make_short2((short)gx, (short)gy);
            }
            short2 *o = out + (size_t)y * w + x0;
            if (vecst) *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(g);
            else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (x0 + j < w) o[j] = g[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { hd2[j] = hd1[j]; hs2[j] = hs1[j]; hd1[j] = hd0[j]; hs1[j] = hs0[j]; }
    }
}

__constant__ int c_lbd_comb[32][2] = {
    { 0, 1 }, { 0, 2 }, { 0, 3 }, { 0, 4 }, { 0, 5 }, { 0, 6 }, { 1, 2 }, { 1, 3 }, { 1, 4 }, { 1, 5 }, { 1, 6 },
    { 2, 3 }, { 2, 4 }, { 2, 5 }, { 2, 6 }, { 2, 7 }, { 2, 8 }, { 3, 4 }, { 3, 5 }, { 3, 6 }, { 3, 7 }, { 3, 8 },
    { 4, 5 }, { 4, 6 }, { 4, 7 }, { 4, 8 }, { 5, 6 }, { 5, 7 }, { 5, 8 }, { 6, 7 }, { 6, 8 }, { 7, 8 } };

#define LBD_UNR 8
__global__ __launch_bounds__(64) void k_lbd_desc(const short2 *__restrict__ dxyImg, int w, int h,
                                                 const hvo_keyline *__restrict__ kls, const int *__restrict__ nkl, int kl_cap,
                                                 const float *__restrict__ gL, const float *__restrict__ gG, uint8_t *__restrict__ desc)
{
    __shared__ float rows[63][4];       // pgdL, ngdL, pgdO, ngdO row sums (already scaled by the global weight)
    __shared__ float band[9][8];
    __shared__ float dv[72];
    const int line = blockIdx.x, f = blockIdx.y, t = threadIdx.x;
    if (line >= nkl[f]) return;
    const hvo_keyline kl = kls[(size_t)f * kl_cap + line];
    const short2 *DXY = dxyImg + (size_t)f * w * h;
    const short imageWidth = (short)(w - 1), imageHeight = (short)(h - 1);
    const short halfHeight = 31;
    const short lengthOfLSP = (short)kl.num_pixels;
    const short halfWidth = (short)((lengthOfLSP - 1) / 2);
    const float midX = (float)(0.5 * (double)__fadd_rn(kl.sox, kl.eox));
    const float midY = (float)(0.5 * (double)__fadd_rn(kl.soy, kl.eoy));
    const float dL0 = (float)cos((double)kl.angle), dL1 = (float)sin((double)kl.angle);
    const float dO0 = -dL1, dO1 = dL0;
    // The samples of row t at positions w0 .. : a sequential chain of float additions per row (lane = row carries it).  What the gathers cost
    // is the number of distinct cache lines a load instruction touches (the texture addresser takes a line per cycle): with lane = row the 63
    // samples of one position lie ACROSS the segment -- in 63 image rows for a segment that runs along the image rows, i.e. 63 lines per load.
    // Such segments (|dL0| > |dL1|) go through an LDS tile (round 5): lane = row forms 32 positions' addresses, the tile is read back with
    // lane = position (two band rows per instruction: 32 consecutive samples ALONG the segment share one or two lines), the values return
    // through the tile and lane = row accumulates them in the reference's order.  Steep segments keep the direct form (their 63 rows lie
    // along an image row already).
    __shared__ int tile[63 * 33];
    const bool along_rows = fabsf(dL0) > fabsf(dL1);
    float sCorX = 0, sCorY = 0, pL = 0, nL = 0, pO = 0, nO = 0;
    if (t < 63) {
        // sCorX0/sCorY0 of row hID are reached by hID sequential updates in the reference
        float sCorX0 = __fadd_rn(__fadd_rn(__fmul_rn(-dL0, (float)halfWidth), __fmul_rn(dL1, (float)halfHeight)), midX);
        float sCorY0 = __fadd_rn(__fsub_rn(__fmul_rn(-dL1, (float)halfWidth), __fmul_rn(dL0, (float)halfHeight)), midY);
        for (int q = 0; q < t; q++) { sCorX0 = __fsub_rn(sCorX0, dL1); sCorY0 = __fadd_rn(sCorY0, dL0); }
        sCorX = sCorX0; sCorY = sCorY0;
    }
    auto accumulate = [&](short gx, short gy) {
        const float ddx = (float)gx, ddy = (float)gy;
        const float gDL = __fadd_rn(__fmul_rn(ddx, dL0), __fmul_rn(ddy, dL1));
        const float gDO = __fadd_rn(__fmul_rn(ddx, dO0), __fmul_rn(ddy, dO1));
        if (gDL > 0) pL = __fadd_rn(pL, gDL); else nL = __fsub_rn(nL, gDL);
        if (gDO > 0) pO = __fadd_rn(pO, gDO); else nO = __fsub_rn(nO, gDO);
    };
    if (along_rows) {
        for (int w0 = 0; w0 < lengthOfLSP; w0 += 32) {
            if (t < 63) {
#pragma unroll 4
                for (int u = 0; u < 32; u++) {
                    int a = -1;
                    if (w0 + u < lengthOfLSP) {
                        short tc = (short)roundf(sCorX);
                        const short xCor = (tc < 0) ? 0 : (tc > imageWidth) ? imageWidth : tc;
                        tc = (short)roundf(sCorY);
                        const short yCor = (tc < 0) ? 0 : (tc > imageHeight) ? imageHeight : tc;
                        a = (int)yCor * w + xCor;
                        sCorX = __fadd_rn(sCorX, dL0); sCorY = __fadd_rn(sCorY, dL1);
                    }
                    tile[t * 33 + u] = a;
                }
            }
            __syncthreads();
            {
                const int u = t & 31, rh = t >> 5;
                for (int r0 = 0; r0 < 64; r0 += 16) {              // eight instructions' loads in flight
                    int av[8]; short2 gv[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { const int r = r0 + 2 * k + rh; av[k] = r < 63 ? tile[r * 33 + u] : -1; }
#pragma unroll
                    for (int k = 0; k < 8; k++) { gv[k] = make_short2(0, 0); if (av[k] >= 0) gv[k] = DXY[av[k]]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) { const int r = r0 + 2 * k + rh; if (r < 63) tile[r * 33 + u] = (int)(((unsigned)(unsigned short)gv[k].x) | ((unsigned)(unsigned short)gv[k].y << 16)); }
                }
            }
            __syncthreads();
            if (t < 63) {
                const int nu = min(32, (int)lengthOfLSP - w0);
                for (int u = 0; u < nu; u++) { const unsigned b = (unsigned)tile[t * 33 + u]; accumulate((short)(b & 0xFFFFu), (short)(b >> 16)); }
            }
            __syncthreads();
        }
    } else if (t < 63) {
        // Samples are taken LBD_UNR at a time: the sample coordinates form a cheap sequential chain, the gathers that
        // depend on them are all issued before the first one is consumed, and the signed sums are then accumulated in
        // the reference's order.
        for (int w0 = 0; w0 < lengthOfLSP; w0 += LBD_UNR) {
            short2 g2[LBD_UNR];
#pragma unroll
            for (int u = 0; u < LBD_UNR; u++) {
                g2[u] = make_short2(0, 0);
                if (w0 + u < lengthOfLSP) {
                    // (short)round((double)v) of the reference: a float's nearest integer (ties away from zero) is the
                    // same whether it is formed in float or in double
                    short tc = (short)roundf(sCorX);
                    const short xCor = (tc < 0) ? 0 : (tc > imageWidth) ? imageWidth : tc;
                    tc = (short)roundf(sCorY);
                    const short yCor = (tc < 0) ? 0 : (tc > imageHeight) ? imageHeight : tc;
                    g2[u] = DXY[(int)yCor * w + xCor];
                    sCorX = __fadd_rn(sCorX, dL0); sCorY = __fadd_rn(sCorY, dL1);
                }
            }
#pragma unroll
            for (int u = 0; u < LBD_UNR; u++) if (w0 + u < lengthOfLSP) accumulate(g2[u].x, g2[u].y);
        }
    }
    if (t < 63) {
        const float coef = gG[t];
        rows[t][0] = __fmul_rn(coef, pL); rows[t][1] = __fmul_rn(coef, nL); rows[t][2] = __fmul_rn(coef, pO); rows[t][3] = __fmul_rn(coef, nO);
    }
    __syncthreads();
    if (t < 9) {
        // band b receives rows 7(b-1)..7(b+1)+6 in row order: "below" (gL[h%7]), own (gL[h%7+7]), "above" (gL[h%7+14])
        float s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int hID = 7 * (t - 1); hID < 7 * (t + 2); hID++) {
            if (hID < 0 || hID >= 63) continue;
            const int rb = hID / 7, m = hID - 7 * rb;
            const float c = gL[rb == t ? m + 7 : (rb == t + 1 ? m + 14 : m)];
            const float pL = rows[hID][0], nL = rows[hID][1], pO = rows[hID][2], nO = rows[hID][3];
            const float cc = __fmul_rn(c, c);
            s[0] = __fadd_rn(s[0], __fmul_rn(c, pL)); s[1] = __fadd_rn(s[1], __fmul_rn(c, nL));
            s[2] = __fadd_rn(s[2], __fmul_rn(cc, __fmul_rn(pL, pL))); s[3] = __fadd_rn(s[3], __fmul_rn(cc, __fmul_rn(nL, nL)));
            s[4] = __fadd_rn(s[4], __fmul_rn(c, pO)); s[5] = __fadd_rn(s[5], __fmul_rn(c, nO));
            s[6] = __fadd_rn(s[6], __fmul_rn(cc, __fmul_rn(pO, pO))); s[7] = __fadd_rn(s[7], __fmul_rn(cc, __fmul_rn(nO, nO)));
        }
        for (int q = 0; q < 8; q++) band[t][q] = s[q];
    }
    __syncthreads();
    // Mean / standard deviation of the nine bands, the three normalisations and the clip (binary_descriptor_custom.cpp:1256-1341).  One lane
    // used to walk all 72 values four times (~1400 wave-instructions of a 3200-instruction kernel that is bound by issue); now lane b < 9 forms
    // band b's eight values, every lane keeps one value (lanes 0..7 a second one: 64 + lane) and scales / clips it, and only the three sums --
    // whose order of additions is the reference's -- are chains: every term is read from its lane (v_readlane, a uniform operand) and added by
    // all lanes at once, so the totals need no broadcast.
    if (t < 9) {
        const float invN = (t == 0 || t == 8) ? (float)(1.0 / 14.0) : (float)(1.0 / 21.0);
        float tmp = __fmul_rn(band[t][0], invN);
        dv[8 * t] = tmp; dv[8 * t + 4] = sqrtf(__fsub_rn(__fmul_rn(band[t][2], invN), __fmul_rn(tmp, tmp)));
        tmp = __fmul_rn(band[t][1], invN);
        dv[8 * t + 1] = tmp; dv[8 * t + 5] = sqrtf(__fsub_rn(__fmul_rn(band[t][3], invN), __fmul_rn(tmp, tmp)));
        tmp = __fmul_rn(band[t][4], invN);
        dv[8 * t + 2] = tmp; dv[8 * t + 6] = sqrtf(__fsub_rn(__fmul_rn(band[t][6], invN), __fmul_rn(tmp, tmp)));
        tmp = __fmul_rn(band[t][5], invN);
        dv[8 * t + 3] = tmp; dv[8 * t + 7] = sqrtf(__fsub_rn(__fmul_rn(band[t][7], invN), __fmul_rn(tmp, tmp)));
    }
    __syncthreads();
    {
        float v0 = dv[t], v1 = t < 8 ? dv[64 + t] : 0.f;           // value t, and value 64 + t on lanes 0..7
        auto lane_of = [](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
        float p0 = __fmul_rn(v0, v0), p1 = __fmul_rn(v1, v1);
        float tempM = 0, tempS = 0;
#pragma unroll
        for (int i = 0; i < 72; i++) {
            const float term = i < 64 ? lane_of(p0, i & 63) : lane_of(p1, i & 7);
            if ((i & 7) < 4) tempM = __fadd_rn(tempM, term); else tempS = __fadd_rn(tempS, term);
        }
        tempM = __fdiv_rn(1.f, sqrtf(tempM)); tempS = __fdiv_rn(1.f, sqrtf(tempS));
        v0 = __fmul_rn(v0, (t & 7) < 4 ? tempM : tempS); v1 = __fmul_rn(v1, (t & 7) < 4 ? tempM : tempS);
        if ((double)v0 > 0.4) v0 = (float)0.4;
        if ((double)v1 > 0.4) v1 = (float)0.4;
        p0 = __fmul_rn(v0, v0); p1 = __fmul_rn(v1, v1);
        float tmp = 0;
#pragma unroll
        for (int i = 0; i < 72; i++) tmp = __fadd_rn(tmp, i < 64 ? lane_of(p0, i & 63) : lane_of(p1, i & 7));
        tmp = __fdiv_rn(1.f, sqrtf(tmp));
        __syncthreads();                                           // every lane has read its value(s)
        dv[t] = __fmul_rn(v0, tmp);
        if (t < 8) dv[64 + t] = __fmul_rn(v1, tmp);
    }
    __syncthreads();
    if (t < 32) {
        const float *f1 = dv + 8 * c_lbd_comb[t][0], *f2 = dv + 8 * c_lbd_comb[t][1];
        unsigned r = 0;
        for (int b = 0; b < 8; b++) if (f1[b] > f2[b]) r |= 1u << b;
        desc[((size_t)f * kl_cap + line) * 32 + t] = (uint8_t)r;
    }
}


// ------------------------------------------------------------------------------------------------
// k_cull_lines: Frame::cullingLine (reference src/Frame.cc:952-1116), one wave per frame.
//   step 1 (pair scan, 956-1027): leaders in ascending index; a leader's candidates j > i are tested 64 at a
//           time, the accepted ones are appended in ascending j (ballot order = the reference's push order)
//   step 2 (1030-1056): every leader folds MergeTwoLines over its members (one leader per lane)
//   step 3 (1062-1092): new KeyLines, stable rank sort by response, class_id = rank, line functions
// The second LBD pass (1094-1096) is a k_lbd_desc launch on the result.
// ------------------------------------------------------------------------------------------------
#define CULL_MAXL 2048         // lines per frame cullingLine holds in LDS (round 5: carved from dynamic LDS by the plan's line quota; 512 and static before)
struct CullArgs {
    const hvo_keyline *kl; const double *fn; const int *nkl; hvo_keyline *tmp; hvo_keyline *kl_out; double *fn_out; int *nkl_out;
    int cap, tmp_stride, w, h; double dis, cos_th, endpoint_dis;
};

static __device__ double cull_point_line_distance(const float *l, float px, float py)                  // Frame.cc:1117-1126
{
    const double x0 = (double)px, y0 = (double)py, x1 = l[0], y1 = l[1], x2 = l[2], y2 = l[3];
    return fabs((y2 - y1) * x0 + (x1 - x2) * y0 + ((x2 * y1) - (x1 * y2))) / sqrt((y2 - y1) * (y2 - y1) + (x1 - x2) * (x1 - x2));
}
static __device__ double cull_two_line_angle(const double *f1, const double *f2)                       // Frame.cc:1127-1140, as written
{
    double v1[3] = { f1[0], f1[1], f1[2] }, v2[3] = { f2[0], f2[1], f2[2] };
    v1[0] /= v1[2]; v1[1] /= v1[2];
    v2[0] /= v2[2]; v2[1] /= v2[2];
    const double a0 = v1[0] / v1[2], a1 = v1[1] / v1[2], b0 = v2[0] / v2[2], b1 = v2[1] / v2[2];
    const double a = a0 * b0 + a1 * b1;
    const double b = sqrt(a0 * a0 + a1 * a1), c = sqrt(b0 * b0 + b1 * b1);
    return fabs(a / (b * c));
}
static __device__ void cull_merge_two_lines(const float *l1, const float *l2, float *out)               // Frame.cc:1141-1202
{
    const double PI = 3.1415926535897932384626433832795;
    const float ax = l1[0], ay = l1[1], bx = l1[2], by = l1[3], cx = l2[0], cy = l2[1], dx = l2[2], dy = l2[3];
    const float dlix = __fsub_rn(bx, ax), dliy = __fsub_rn(by, ay), dljx = __fsub_rn(dx, cx), dljy = __fsub_rn(dy, cy);
    const double li = sqrt((double)__fmul_rn(dlix, dlix) + (double)__fmul_rn(dliy, dliy));
    const double lj = sqrt((double)__fmul_rn(dljx, dljx) + (double)__fmul_rn(dljy, dljy));
    const double xg = (li * (double)__fadd_rn(ax, bx) + lj * (double)__fadd_rn(cx, dx)) / (double)(2.0 * (li + lj));
    const double yg = (li * (double)__fadd_rn(ay, by) + lj * (double)__fadd_rn(cy, dy)) / (double)(2.0 * (li + lj));
    const double thi = dlix == 0.0f ? PI / 2.0 : atan((double)__fdiv_rn(dliy, dlix));
    const double thj = dljx == 0.0f ? PI / 2.0 : atan((double)__fdiv_rn(dljy, dljx));
    double thr;
    if (fabs(thi - thj) <= PI / 2.0) thr = (li * thi + lj * thj) / (li + lj);
    else { const double tmp = thj - PI * (thj / fabs(thj)); thr = li * thi + lj * tmp; thr /= (li + lj); }
    const double s = sin(thr), c = cos(thr);
    const double axg = ((double)ay - yg) * s + ((double)ax - xg) * c, bxg = ((double)by - yg) * s + ((double)bx - xg) * c;
    const double cxg = ((double)cy - yg) * s + ((double)cx - xg) * c, dxg = ((double)dy - yg) * s + ((double)dx - xg) * c;
    const double d1 = fmin(axg, fmin(bxg, fmin(cxg, dxg))), d2 = fmax(axg, fmax(bxg, fmax(cxg, dxg)));
    out[0] = (float)(d1 * c + xg); out[1] = (float)(d1 * s + yg); out[2] = (float)(d2 * c + xg); out[3] = (float)(d2 * s + yg);
}
static size_t cull_lds_bytes(int maxl) { return (size_t)maxl * (24 + 16 + 16 + 4 + 2 + 2 + 1) + 64; }
__global__ __launch_bounds__(64) void k_cull_lines(CullArgs a, int maxl)
{
    // 65 bytes of LDS per line, for `maxl` lines (the plan's quota rounded up to 64): doubles first
    extern __shared__ __attribute__((aligned(16))) unsigned char cull_lds[];
    double (*fnl)[3] = reinterpret_cast<double (*)[3]>(cull_lds);
    float (*ep)[4] = reinterpret_cast<float (*)[4]>(cull_lds + (size_t)maxl * 24);              // end points of the input lines
    float (*nl)[4] = reinterpret_cast<float (*)[4]>(cull_lds + (size_t)maxl * 40);              // merged / surviving segments
    float *resp = reinterpret_cast<float *>(cull_lds + (size_t)maxl * 56);
    short *grp = reinterpret_cast<short *>(cull_lds + (size_t)maxl * 60);
    short *gstart = reinterpret_cast<short *>(cull_lds + (size_t)maxl * 62);                    // maxl + 1 entries (the slack behind the last array holds the last one)
    unsigned char *tag = cull_lds + (size_t)maxl * 64 + 8;
    const int f = blockIdx.x, lane = threadIdx.x;
    const int n = min(a.nkl[f], maxl);
    const hvo_keyline *kl = a.kl + (size_t)f * a.cap;
    const double *fn = a.fn + (size_t)f * a.cap * 3;
    for (int i = lane; i < n; i += 64) {
        ep[i][0] = kl[i].sx; ep[i][1] = kl[i].sy; ep[i][2] = kl[i].ex; ep[i][3] = kl[i].ey;
        fnl[i][0] = fn[3 * i]; fnl[i][1] = fn[3 * i + 1]; fnl[i][2] = fn[3 * i + 2];
        tag[i] = 0;
    }
    __syncthreads();
    // ---- step 1 ----
    int ngrp = 0;
    for (int i = 0; i < n; i++) {
        if (lane == 0) gstart[i] = (short)ngrp;
        if (tag[i]) continue;                     // uniform (LDS broadcast)
        const float *e1 = ep[i];
        const float m12x = (float)((double)__fadd_rn(e1[0], e1[2]) * 0.5), m12y = (float)((double)__fadd_rn(e1[1], e1[3]) * 0.5);
        bool any = false;
        for (int base = i + 1; base < n; base += 64) {
            const int j = base + lane;
            bool acc = false;
            if (j < n && !tag[j]) {
                const float *e2 = ep[j];
                float m21x = (float)((double)__fadd_rn(e2[2], e2[0]) * 0.5), m21y = (float)((double)__fadd_rn(e2[3], e2[1]) * 0.5);
                m21x = __fadd_rn(m21x, e2[0]); m21y = __fadd_rn(m21y, e2[1]);                  // Frame.cc:977, as written
                const double dis12 = cull_point_line_distance(e2, m12x, m12y), dis21 = cull_point_line_distance(e1, m21x, m21y);
                if ((dis12 < a.dis || dis21 < a.dis) && cull_two_line_angle(fnl[i], fnl[j]) > a.cos_th) {
                    double bx[4] = { (double)e1[0], (double)e1[2], (double)e2[0], (double)e2[2] }, by[4] = { (double)e1[1], (double)e1[3], (double)e2[1], (double)e2[3] };
#pragma unroll
                    for (int p = 1; p < 4; p++) for (int q = p; q > 0; q--) {
                        if (bx[q - 1] > bx[q]) { const double t = bx[q]; bx[q] = bx[q - 1]; bx[q - 1] = t; }
                        if (by[q - 1] > by[q]) { const double t = by[q]; by[q] = by[q - 1]; by[q - 1] = t; }
                    }
                    const double dx = bx[3] - bx[0], dy = by[3] - by[0];
                    const double dx1 = fabs((double)e1[0] - (double)e1[2]), dx2 = fabs((double)e2[0] - (double)e2[2]);
                    const double dy1 = fabs((double)e1[1] - (double)e1[3]), dy2 = fabs((double)e2[1] - (double)e2[3]);
                    acc = !(dx > dx1 + dx2 && bx[2] - bx[1] > a.endpoint_dis) && !(dy > dy1 + dy2 && by[2] - by[1] > a.endpoint_dis);
                }
            }
            const unsigned long long m = __ballot(acc);
            if (acc) { grp[ngrp + __popcll(m & ((1ull << lane) - 1))] = (short)j; tag[j] = 1; }
            ngrp += __popcll(m);
            any |= m != 0;
        }
        if (any && lane == 0) tag[i] = 1;
        __syncthreads();
    }
    if (lane == 0) gstart[n] = (short)ngrp;
    __syncthreads();
    // ---- step 2: one leader per lane ----
    int m = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool keep = false; float cur[4] = { 0, 0, 0, 0 };
        if (i < n) {
            const int g0 = gstart[i], g1 = gstart[i + 1];
            cur[0] = ep[i][0]; cur[1] = ep[i][1]; cur[2] = ep[i][2]; cur[3] = ep[i][3];
            for (int q = g0; q < g1; q++) { float r[4]; cull_merge_two_lines(cur, ep[grp[q]], r); cur[0] = r[0]; cur[1] = r[1]; cur[2] = r[2]; cur[3] = r[3]; }
            keep = g1 > g0 || !tag[i];
        }
        const unsigned long long km = __ballot(keep);
        if (keep) { const int p = m + __popcll(km & ((1ull << lane) - 1)); nl[p][0] = cur[0]; nl[p][1] = cur[1]; nl[p][2] = cur[2]; nl[p][3] = cur[3]; }
        m += __popcll(km);
    }
    __syncthreads();
    // ---- step 3: KeyLines, stable rank sort by response, line functions ----
    hvo_keyline *tmp = a.tmp + (size_t)f * a.tmp_stride;
    for (int i = lane; i < m; i += 64) {
        hvo_keyline k;
        k.sx = k.sox = nl[i][0]; k.sy = k.soy = nl[i][1]; k.ex = k.eox = nl[i][2]; k.ey = k.eoy = nl[i][3];
        const double ddx = (double)__fsub_rn(nl[i][0], nl[i][2]), ddy = (double)__fsub_rn(nl[i][1], nl[i][3]);
        k.length = (float)sqrt(ddx * ddx + ddy * ddy);
        k.octave = 0;
        k.angle = (float)atan2((double)__fsub_rn(k.ey, k.sy), (double)__fsub_rn(k.ex, k.sx));
        k.size = __fmul_rn(__fsub_rn(k.ex, k.sx), __fsub_rn(k.ey, k.sy));
        k.pt_x = __fdiv_rn(__fadd_rn(k.ex, k.sx), 2.f); k.pt_y = __fdiv_rn(__fadd_rn(k.ey, k.sy), 2.f);
        k.num_pixels = cull_line_count(a.w, a.h, nl[i][0], nl[i][1], nl[i][2], nl[i][3]);
        k.response = __fdiv_rn(k.length, (float)(a.w > a.h ? a.w : a.h));
        k.class_id = -1;
        tmp[i] = k; resp[i] = k.response;
    }
    __syncthreads();
    hvo_keyline *out = a.kl_out + (size_t)f * a.cap;
    double *fo = a.fn_out + (size_t)f * a.cap * 3;
    for (int i = lane; i < m; i += 64) {
        const float r = resp[i];
        int rank = 0;
        for (int q = 0; q < m; q++) { const float rq = resp[q]; rank += (rq > r) || (rq == r && q < i); }
        hvo_keyline k = tmp[i]; k.class_id = rank;
        if (rank < a.cap) {
            out[rank] = k;
            const double sx = k.sx, sy = k.sy, ex = k.ex, ey = k.ey;
            const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
            const double nrm = sqrt(l0 * l0 + l1 * l1);
            fo[3 * rank] = l0 / nrm; fo[3 * rank + 1] = l1 / nrm; fo[3 * rank + 2] = l2 / nrm;
        }
    }
    if (lane == 0) a.nkl_out[f] = m;
}

// ================================================================================================
// host side
// ================================================================================================
void lsd_free(hvo_ctx *ctx)
{
    LsdPlan *P = plan_of(ctx);
    if (!P) return;
    void *ptrs[] = { P->d_kl2, P->d_desc2, P->d_fn2, P->d_nkl2, P->d_blur, P->d_px, P->d_defined, P->d_reg, P->d_segs, P->d_kl_all, P->d_kl,
                     P->d_desc, P->d_fn, P->d_nkl, P->d_flags, P->d_b5, P->d_dxy, P->d_xofs, P->d_yofs, P->d_xa, P->d_yb, P->d_gL, P->d_gG, P->d_stats,
                     P->d_pool, P->d_defmask, P->d_wprefix, P->d_fbase, P->d_fcount, P->d_pooltop, P->d_atags, P->d_actl, P->d_alists, P->d_ablk, P->d_afreg, P->d_ainreg, P->d_redo, P->d_b8, P->d_s8, P->d_rs8tab };
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete P;
    ctx->lsd = nullptr;
}

static int cvfloor_f(float v) { int i = (int)v; return i - (i > v); }

static int lsd_build_plan(hvo_ctx *ctx, int w, int h, int batch);
// as orb_ensure_plan: a plan that fails half-way is freed, so its (w, h, batch) key never outlives its slabs
static int lsd_ensure_plan(hvo_ctx *ctx, int w, int h, int batch)
{
    LsdPlan *P = plan_of(ctx);
    if (P && P->w == w && P->h == h && P->batch >= batch) return HVO_OK;
    lsd_free(ctx);
    const int rc = lsd_build_plan(ctx, w, h, batch);
    if (rc) lsd_free(ctx);
    return rc;
}

static int lsd_build_plan(hvo_ctx *ctx, int w, int h, int batch)
{
    LsdPlan *P = new LsdPlan();
    ctx->lsd = P;
    P->w = w; P->h = h; P->batch = batch; P->nfeat = std::max(ctx->p.lsd_nfeatures, 1);
    P->sw = (int)lrint(w * 0.8); P->sh = (int)lrint(h * 0.8);
    if (P->sw < 8 || P->sh < 8 || w > 32767 || h > 32767) return HVO_ERR_UNSUPPORTED;
    P->nwords = ((P->sw + 31) / 32) * P->sh;
    // LSD constants (OpenCV defaults, LSD_REFINE_STD)
    const double SCALE = 0.8, ANG_TH = 22.5;
    P->prec = LSD_PI * ANG_TH / 180; P->p = ANG_TH / 180; P->rho = 2.0 / sin(P->prec); P->rhoT = lsd_sqrt_threshold(P->rho);
    const double LOG_NT = 5 * (log10((double)P->sw) + log10((double)P->sh)) / 2 + log10(11.0);
    P->min_reg = (unsigned)(-LOG_NT / log10(P->p));
    P->maxseg = (int)(((size_t)P->sw * P->sh) / std::max(P->min_reg, 1u)) + 64;     // every segment is a region of at least min_reg pixels of its own
    {   // getGaussianKernel(7, 0.75, CV_64F)
        const double sigma = 0.6 / SCALE; double k[7], sum = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3.0; k[i] = exp(-0.5 / (sigma * sigma) * x * x); sum += k[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 4; i++) P->k7[i] = k[i] * sum;
    }
    {   // getGaussianKernel(5, 1, CV_32F) * 256, rounded (fixed-point CV_8U path)
        float cf[5]; double sum = 0;
        for (int i = 0; i < 5; i++) { double x = i - 2.0; cf[i] = (float)exp(-0.5 * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 3; i++) P->k5[i] = (int)lrintf((float)(cf[i] * sum) * 256.f);
    }
    {   // BinaryDescriptor weights with the reference's integer divisions (binary_descriptor_custom.cpp:227-258)
        double u = (7 * 3 - 1) / 2, sigma = (7 * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < 21; i++) { double d = i - u; P->gL[i] = exp(d * d * inv); }
        u = (9 * 7 - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < 63; i++) { double d = i - u; P->gG[i] = exp(d * d * inv); }
    }
    std::vector<int> xofs(P->sw), yofs(P->sh); std::vector<float> xa(2 * P->sw), yb(2 * P->sh);
    const double scale = 1. / SCALE;
    for (int dx = 0; dx < P->sw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5); int sx = cvfloor_f(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= w - 1) { fx = 0; sx = w - 1; }
        xofs[dx] = sx; xa[2 * dx] = 1.f - fx; xa[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < P->sh; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5); int sy = cvfloor_f(fy); fy -= sy;
        int sy0 = std::min(std::max(sy, 0), h - 1), sy1 = std::min(std::max(sy + 1, 0), h - 1);
        yofs[dy] = sy0 | (sy1 << 16); yb[2 * dy] = 1.f - fy; yb[2 * dy + 1] = fy;
    }
    std::vector<float> gLf(21), gGf(63);
    for (int i = 0; i < 21; i++) gLf[i] = (float)P->gL[i];
    for (int i = 0; i < 63; i++) gGf[i] = (float)P->gG[i];
    const size_t B = batch, npix = (size_t)w * h, nsp = (size_t)P->sw * P->sh;
#define PA(ptr, n) HVO_HIP(hipMalloc((void **)&(ptr), (n)))
    // transient images (the fp64 blurred image between k_lsd_blur and k_lsd_resize_grad, the Sobel image between k_lbd_blur_sobel and
    // k_lbd_desc) exist for a CHUNK of the batch only: lsd_run walks the batch chunk by chunk through those kernels (3.7 MB per frame saved)
    P->kn.dense.read("HVO_LSD_DENSE"); P->kn.lat.read("HVO_LSD_LAT"); P->kn.async_w.read("HVO_LSD_ASYNC"); P->kn.async_early.read("HVO_LSD_ASYNC_EARLY");
    P->kn.async_lds.read("HVO_LSD_ASYNC_LDS"); P->kn.lat_lds.read("HVO_LSD_LAT_LDS"); P->kn.lbd_split.read("HVO_LBD_SPLIT"); P->kn.spin_max.read("HVO_LSD_ASYNC_SPIN_MAX");
    P->chunk = (int)std::min<size_t>(B, 512);
    { const char *e = getenv("HVO_LSD_CHUNK"); if (e && atoi(e) > 0) P->chunk = (int)std::min<size_t>(B, (size_t)atoi(e)); }
    // compact records (HVO_LSD_COMPACT): the dense record image exists for a chunk only
    // (at most ~3.3 GB of it), the batch keeps a pool sized for HVO_LSD_COMPACT_FRAC of the pixels (lsd_run grows it when a batch asks for more)
    // Opt-in (HVO_LSD_COMPACT=1): it buys memory, not time.  Measured: 22.1 -> 18.0 MB per 640x480 frame for 5 % of the step (the compaction
    // pass and the longer index arithmetic of an issue-bound kernel); 1280x960: 78 -> ~60 MB, 4096 frames resident instead of 3072, but
    // 8.5 k frames/s at 4096 against 8.95 k at 3072 with one record per pixel -- the wave slots are full by then.
    P->compact = false;
    { const char *e = getenv("HVO_LSD_COMPACT"); if (e) P->compact = atoi(e) != 0; }
    double frac = 0.27;
    { const char *e = getenv("HVO_LSD_COMPACT_FRAC"); if (e && atof(e) > 0) frac = std::min(1.0, atof(e)); }
    if (P->compact) {
        const size_t fit = std::max<size_t>(64, (size_t)(3.3e9 / (double)(nsp * sizeof(double4))) / 64 * 64);
        P->chunk = (int)std::min<size_t>((size_t)P->chunk, std::min<size_t>(fit, 512));
    }
    const size_t CB = (size_t)P->chunk;
    { const char *e = getenv("HVO_LSD_PRE_SPLIT"); P->pre_fused = !(e && atoi(e) != 0); }
    // the tile of k_lsd_pre holds the source rectangle of PRE_TW x PRE_TR scaled pixels: true for the 0.8 scale of every geometry (asserted here)
    if (!P->pre_fused) PA(P->d_blur, CB * npix * 8);
    if (P->compact) {
        PA(P->d_px, CB * nsp * sizeof(double4));
        P->pool_cap = std::max<size_t>((size_t)((double)(B * nsp) * frac), 4096);
        PA(P->d_pool, P->pool_cap * sizeof(double4));
        PA(P->d_defmask, B * P->nwords * 4); PA(P->d_wprefix, B * P->nwords * 4);
        PA(P->d_fbase, B * 8); PA(P->d_fcount, B * 4); PA(P->d_pooltop, 16);
    } else
        PA(P->d_px, B * nsp * sizeof(double4));
    PA(P->d_defined, B * P->nwords * 4); PA(P->d_reg, B * nsp * 4);
    PA(P->d_segs, B * P->maxseg * 16); PA(P->d_kl_all, B * P->maxseg * sizeof(hvo_keyline));
    PA(P->d_kl, B * P->nfeat * sizeof(hvo_keyline)); PA(P->d_desc, B * P->nfeat * 32); PA(P->d_fn, B * P->nfeat * 24);
    PA(P->d_nkl, B * 4); PA(P->d_flags, B * 4);
    PA(P->d_kl2, B * P->nfeat * sizeof(hvo_keyline)); PA(P->d_desc2, B * P->nfeat * 32); PA(P->d_fn2, B * P->nfeat * 24); PA(P->d_nkl2, B * 4);
    if (P->kn.lbd_split.set) PA(P->d_b5, CB * npix);
    PA(P->d_dxy, CB * npix * sizeof(short2));
    PA(P->d_xofs, P->sw * 4); PA(P->d_yofs, P->sh * 4); PA(P->d_xa, P->sw * 8); PA(P->d_yb, P->sh * 8);
    PA(P->d_gL, 21 * 4); PA(P->d_gG, 63 * 4); PA(P->d_stats, B * 64);
    if (P->pre_fused) {
        // a band's source columns must fit the workgroup's 256 threads (true for the 0.8x scale: 192 / 0.8 + 2; checked, not assumed)
        int maxc = 0;
        for (int x0 = 0; x0 < P->sw; x0 += PRE_TW) { const int xe = std::min(x0 + PRE_TW, P->sw - 1); maxc = std::max(maxc, std::min(xofs[xe] + 1, w - 1) - xofs[x0] + 1); }
        if (maxc > 256) { P->pre_fused = false; PA(P->d_blur, CB * npix * 8); }
    }
#undef PA
    HVO_HIP(hipMemcpy(P->d_xofs, xofs.data(), P->sw * 4, hipMemcpyHostToDevice));
    HVO_HIP(hipMemcpy(P->d_yofs, yofs.data(), P->sh * 4, hipMemcpyHostToDevice));
    HVO_HIP(hipMemcpy(P->d_xa, xa.data(), P->sw * 8, hipMemcpyHostToDevice));
    HVO_HIP(hipMemcpy(P->d_yb, yb.data(), P->sh * 8, hipMemcpyHostToDevice));
    HVO_HIP(hipMemcpy(P->d_gL, gLf.data(), 21 * 4, hipMemcpyHostToDevice));
    HVO_HIP(hipMemcpy(P->d_gG, gGf.data(), 63 * 4, hipMemcpyHostToDevice));
    HVO_HIP(hipMemsetAsync(P->d_defined, 0, B * P->nwords * 4, ctx->s_lsd));
    HVO_HIP(hipMemsetAsync(P->d_nkl, 0, B * 4, ctx->s_lsd));
    HVO_HIP(hipMemsetAsync(P->d_nkl2, 0, B * 4, ctx->s_lsd));
    HVO_HIP(hipMemsetAsync(P->d_flags, 0, B * 4, ctx->s_lsd));
    HVO_HIP(hipDeviceSynchronize());
    return HVO_OK;
}

int lsd_prepare(hvo_ctx *ctx, int w, int h, int batch, bool culled, LsdView *v)
{
    int rc = lsd_ensure_plan(ctx, w, h, batch);
    if (rc) return rc;
    LsdPlan *P = plan_of(ctx);
    v->d_kl = culled ? P->d_kl2 : P->d_kl; v->d_desc = culled ? P->d_desc2 : P->d_desc; v->d_fn = culled ? P->d_fn2 : P->d_fn;
    v->d_nkl = culled ? P->d_nkl2 : P->d_nkl; v->d_flags = P->d_flags; v->nfeat = P->nfeat;
    return HVO_OK;
}

int lsd_run(hvo_ctx *ctx, int n, bool cull)
{
    // input: level 0 of the ORB pyramid slab (the uploaded gray image)
    OrbPlan &O = ctx->orb;
    if (O.w <= 0 || n < 1 || n > O.batch) return HVO_ERR_INVALID_ARG;
    int rc = lsd_ensure_plan(ctx, O.w, O.h, std::max(n, ctx->p.max_batch));
    if (rc) return rc;
    LsdPlan *P = plan_of(ctx);
    hipStream_t st = hvo_stream_lsd(ctx);
    const int w = P->w, h = P->h, sw = P->sw, sh = P->sh;
    const uint8_t *gray = O.d_pyr + O.lev[0].img_off;
    const int gpitch = O.lev[0].pitch;
    int id;
    const size_t nsp = (size_t)sw * sh;
    const int gx = (((sw + 31) & ~31) + 255) / 256;
    for (int attempt = 0; attempt < 3; attempt++) {
        if (P->compact) HVO_HIP(hipMemsetAsync(P->d_pooltop, 0, 16, st));
        for (int c0 = 0; c0 < n; c0 += P->chunk) {
            const int m = std::min(P->chunk, n - c0);
            if (ctx->readings & HVO_READING_LSD_8U) {
                // the detector on CV_8U (readings.hip): u8 GaussianBlur 7 x 7 sigma 0.75, u8 resize 0.8x, gradients of the rounded bytes
                id = hvo_prof_begin(ctx, "lsd_gradient", st);
                const size_t npix_ = (size_t)w * h;
                if (!P->d_b8) {
                    HVO_HIP(hipMalloc((void **)&P->d_b8, (size_t)P->chunk * npix_)); HVO_HIP(hipMalloc((void **)&P->d_s8, (size_t)P->chunk * nsp));
                    HVO_HIP(hipMalloc((void **)&P->d_rs8tab, (2 * (size_t)sw + 2 * (size_t)sh) * sizeof(int)));
                    if ((rc = readings_resize_tables(st, w, h, sw, sh, 0.8, P->d_rs8tab))) return rc;
                }
                if ((rc = readings_gblur_enqueue(st, gray + (size_t)c0 * O.pyr_bytes, O.pyr_bytes, gpitch, w, h, P->d_b8, npix_, w, m, 7, 0.6 / 0.8, (ctx->readings & HVO_READING_BLUR_FLOAT) != 0))) return rc;
                if ((rc = readings_resize_enqueue(st, P->d_b8, npix_, w, w, P->d_s8, nsp, sw, sw, sh, m, P->d_rs8tab))) return rc;
                if ((rc = readings_lsd_grad8_enqueue(st, P->d_s8, nsp, sw, sh, P->d_px + (P->compact ? 0 : (size_t)c0 * nsp), P->d_defined + (size_t)c0 * P->nwords, P->nwords, P->rho, m))) return rc;
            } else
            if (P->pre_fused) {
                // one kernel from the u8 image to the records (k_lsd_pre); the fp64 blurred image is never written
                id = hvo_prof_begin(ctx, "lsd_gradient", st);
                hipLaunchKernelGGL(k_lsd_pre, dim3((sw + PRE_TW - 1) / PRE_TW, (sh - 1 + PRE_SEG - 1) / PRE_SEG, m), dim3(256), 0, st, gray + (size_t)c0 * O.pyr_bytes, O.pyr_bytes, gpitch,
                                   w, h, sw, sh, P->d_xofs, P->d_xa, P->d_yofs, P->d_yb, P->d_px + (P->compact ? 0 : (size_t)c0 * nsp),
                                   P->d_defined + (size_t)c0 * P->nwords, P->nwords, P->rhoT, P->k7[0], P->k7[1], P->k7[2], P->k7[3], (const int *)nullptr);
            } else {
            id = hvo_prof_begin(ctx, "lsd_blur_scale", st);
            hipLaunchKernelGGL(k_lsd_blur, dim3((w + 255) / 256, (h + LSD_BLUR_ROWS - 1) / LSD_BLUR_ROWS, m), dim3(256), 0, st, gray + (size_t)c0 * O.pyr_bytes, O.pyr_bytes, gpitch, P->d_blur, w, h,
                               P->k7[0], P->k7[1], P->k7[2], P->k7[3]);
            hvo_prof_end(ctx, id);
            id = hvo_prof_begin(ctx, "lsd_gradient", st);
            hipLaunchKernelGGL(k_lsd_resize_grad, dim3(gx, (sh + GRAD_ROWS - 1) / GRAD_ROWS, m), dim3(256), 0, st, P->d_blur, w, h, sw, sh, P->d_xofs, P->d_xa, P->d_yofs, P->d_yb,
                               P->d_px + (P->compact ? 0 : (size_t)c0 * nsp), P->d_defined + (size_t)c0 * P->nwords, P->nwords, P->rho);
            }
            if (P->compact) {
                hipLaunchKernelGGL(k_lsd_prefix, dim3(m), dim3(256), 0, st, P->d_defined + (size_t)c0 * P->nwords, P->d_defmask + (size_t)c0 * P->nwords,
                                   P->d_wprefix + (size_t)c0 * P->nwords, P->d_fcount + c0, P->nwords);
                hipLaunchKernelGGL(k_lsd_bases, dim3(1), dim3(512), 0, st, P->d_fcount + c0, m, P->d_fbase + c0, P->d_pooltop, (unsigned long long)P->pool_cap);
                hipLaunchKernelGGL(k_lsd_compact, dim3(std::min((P->nwords + 255) / 256, 16), m), dim3(256), 0, st, P->d_px, nsp, P->d_defmask + (size_t)c0 * P->nwords,
                                   P->d_wprefix + (size_t)c0 * P->nwords, P->d_fbase + c0, P->d_pool, P->nwords, sw, (sw + 31) / 32);
            }
            hvo_prof_end(ctx, id);
        }
        if (!P->compact) break;
        // Does the pool hold what the batch asked for?  One host wait on this stream (the other stages are enqueued and running; the growing
        // kernel waits for FAST under the large-batch policies anyway).  If not: a larger pool, and the preamble once more.
        unsigned long long need = 0;
        HVO_HIP(hipMemcpyAsync(&need, P->d_pooltop, 8, hipMemcpyDeviceToHost, st));
        HVO_HIP(hipStreamSynchronize(st));
        if (need <= P->pool_cap) break;
        if (attempt == 2) return HVO_ERR_CAPACITY;
        // the larger pool first: a failed allocation must not leave a null pool behind an enlarged capacity (the next run would pass the
        // capacity test and the kernels would dereference it)
        const size_t new_cap = (size_t)((double)need * 1.1) + 4096;
        double4 *np_ = nullptr;
        if (hipMalloc((void **)&np_, new_cap * sizeof(double4)) != hipSuccess) { (void)hipGetLastError(); ctx->last_error = "LSD record pool: out of device memory"; return HVO_ERR_HIP; }
        HVO_HIP(hipFree(P->d_pool));
        P->d_pool = np_; P->pool_cap = new_cap;
    }
    if (ctx->ev_lsd_pre && !ctx->serialize) { HVO_HIP(hipEventRecord(ctx->ev_lsd_pre, st)); ctx->lsd_pre_recorded = true; }
    if ((ctx->sched == 2 || ctx->sched == 5 || ctx->sched == 6) && ctx->fast_recorded && !ctx->serialize) HVO_HIP(hipStreamWaitEvent(st, ctx->ev_fast, 0));
    id = hvo_prof_begin(ctx, "lsd_grow", st);
    GrowArgs g;
    g.px4 = P->compact ? P->d_pool : P->d_px; g.avail = P->d_defined; g.reg = P->d_reg; g.segs = P->d_segs; g.perm = hvo_frame_perm(ctx, n);
    g.defmask = P->d_defmask; g.wprefix = P->d_wprefix; g.fbase = P->d_fbase;
    g.stats = P->d_stats; g.kl_all = P->d_kl_all; g.kl = P->d_kl; g.fn = P->d_fn; g.nkl = P->d_nkl; g.flags = P->d_flags;
    g.sw = sw; g.sh = sh; g.nwords = P->nwords; g.w = w; g.h = h; g.nfeat = P->nfeat; g.kl_cap = P->nfeat;
    g.rho = P->rho; g.prec = P->prec; g.p = P->p; g.min_reg = P->min_reg; g.maxseg = P->maxseg; g.redo = 0; g.redo_count = nullptr;
    bool dense = n > 5 * 1024;
    if (P->kn.dense.set) dense = P->kn.dense.v != 0;              // HVO_LSD_DENSE: tests force either kernel on small batches
    // a handful of frames (the streamed mode, a tracker's small batches): the latency variant with the mask in LDS
    bool lat = n <= 64 && (size_t)P->nwords * 4 <= 150 * 1024 && !P->compact;
    if (P->kn.lat.set) lat = P->kn.lat.v != 0 && (size_t)P->nwords * 4 <= 150 * 1024 && !P->compact;      // HVO_LSD_LAT
    // a handful of frames: W waves per frame grow regions side by side and commit them in seed order (lsd_async.inc); HVO_LSD_ASYNC = W, 0: off
    // (default: up to 16 frames; at 32 frames the one-wave kernel beside the plane chain is the faster whole, tools/latency.py)
    // Workers per frame.  A handful of frames: as many as pay (32 / 16; 64 for a lone large frame).  Round 5 lifted the 16-frame limit of the
    // scratch (allocated per (frames, W) now: HVO_LSD_ASYNC = W works for up to 1024 frames) and MEASURED 64-512 frames with W = 2-16
    // (profiles/r05_async_midsize_batches.txt): slower than one wave per frame everywhere but at 128 frames (-7 %) -- batch256 26.5 -> 31.3 ms
    // at W = 4, 33.5 at W = 8, 34.8 at W = 2 -- the workers that wait for their turn poll, and with a wave on every SIMD already their polling
    // takes the issue slots the growing workers need.  The default stays: async up to 16 frames.
    int aw = n <= 8 ? 32 : n <= 16 ? 16 : 0;
    if (n <= 2 && (size_t)sw * sh >= 600000) aw = 64;          // a lone large frame (1280x960: 5.4 k seeds): more regions in flight
    if (P->kn.async_w.set) aw = std::min(std::max(P->kn.async_w.v, 0), LA_MAXW);
    if (ctx->readings & HVO_READING_LSD_8U) aw = 0;             // (the async growing's fall-back forms a frame again with the default preamble)
    if (aw > 0 && n <= 1024 && !P->compact) {
        // scratch for (n frames, aw workers): owner tags and control block per frame; region list, held-pixel list and membership byte map per worker
        const size_t need_f = (size_t)n, need_w = (size_t)n * aw;
        if (need_f > (size_t)P->async_b || need_w > P->async_w_cap) {
            const size_t AB = std::max(need_f, (size_t)P->async_b), AW = std::max(need_w, P->async_w_cap);
            HVO_HIP(hipStreamSynchronize(st));
            for (void *q : { (void *)P->d_atags, (void *)P->d_actl, (void *)P->d_alists, (void *)P->d_ablk, (void *)P->d_afreg, (void *)P->d_ainreg }) if (q) (void)hipFree(q);
            P->d_atags = nullptr; P->d_actl = nullptr; P->d_alists = nullptr; P->d_ablk = nullptr; P->d_afreg = nullptr; P->d_ainreg = nullptr; P->async_b = 0; P->async_w_cap = 0;
            HVO_HIP(hipMalloc((void **)&P->d_atags, AB * P->nwords * 32 * 4)); HVO_HIP(hipMalloc((void **)&P->d_actl, AB * sizeof(LaCtl)));
            HVO_HIP(hipMalloc((void **)&P->d_alists, AW * LA_CAP * 4)); HVO_HIP(hipMalloc((void **)&P->d_ablk, AW * 2 * LA_BCAP * 4));
            HVO_HIP(hipMalloc((void **)&P->d_afreg, AB * 2 * nsp * 4));
            HVO_HIP(hipMalloc((void **)&P->d_ainreg, AW * P->nwords * 32));            // a worker's map of its region, a byte per scaled pixel
            if (!P->d_redo) HVO_HIP(hipMalloc((void **)&P->d_redo, 64));
            P->async_b = (int)AB; P->async_w_cap = AW; P->ainreg_b = (int)AB;
        }
        {
            HVO_HIP(hipMemsetAsync(P->d_actl, 0, (size_t)n * sizeof(LaCtl), st));
            HVO_HIP(hipMemsetAsync(P->d_redo, 0, 4, st)); P->async_last_n = n; P->async_last_w = aw;
            // tags and region bitmaps are all-free / all-zero after a launch that ran to its end; a launch that aborted (flag 4) may have
            // left some behind, so every launch starts from a clean state (7 MB per 640x480 frame at 32 workers, microseconds)
            HVO_HIP(hipMemsetAsync(P->d_atags, 0xFF, (size_t)n * P->nwords * 32 * 4, st));
            HVO_HIP(hipMemsetAsync(P->d_ainreg, 0, (size_t)n * aw * P->nwords * 32, st));
            LaArgs a; a.g = g; a.tags = P->d_atags; a.inreg = P->d_ainreg; a.ctl = (LaCtl *)P->d_actl; a.lists = P->d_alists; a.blocked = P->d_ablk; a.freg = P->d_afreg; a.W = aw; a.n = n; a.early = P->kn.async_early.or_(1);
            // the turn's wait is bounded (a frame whose wait expires is grown again below, by the one-wave kernel); HVO_LSD_ASYNC_SPIN_MAX: a test hook
            a.spin_max = P->kn.spin_max.set ? (unsigned)std::max(P->kn.spin_max.v, 0) : LA_SPIN_MAX;
            // (an LDS request keeps these one-wave workgroups off the CUs where a frame's AHC waves sit -- k_peac_cluster_heads takes 108 KB --:
            // both are bound by instruction issue and a shared SIMD slows both; HVO_LSD_ASYNC_LDS: bytes [0 for a lone frame's 32-64 workers, 56 K beside other frames])
            size_t alds = (n > 2 && n <= 16) ? 56 * 1024 : 0;       // (more frames: the workers ARE the machine's load, nothing to keep them away from)
            if (P->kn.async_lds.set) alds = (size_t)std::min(std::max(P->kn.async_lds.v, 0), 150 * 1024);
            if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_lsd_grow_async), alds)) return HVO_ERR_HIP;
            hipLaunchKernelGGL(k_lsd_grow_async, dim3(((n + 7) / 8) * 8 * aw), dim3(64), alds, st, a);      // workgroups b, b + 8, ... of a frame: one XCD
            // FAIL SOFT: a frame the workers gave up on (flag 4: the bounded wait expired, or no worker sat on the frame's XCD) is
            // answered with lines, not with an error -- its records and availability mask are formed again (the commits consumed the mask) and
            // the deterministic one-wave kernel grows it; both launches return at once for every other frame (~10 us per call when nothing failed)
            if (P->pre_fused)
                hipLaunchKernelGGL(k_lsd_pre, dim3((sw + PRE_TW - 1) / PRE_TW, (sh - 1 + PRE_SEG - 1) / PRE_SEG, n), dim3(256), 0, st, gray, O.pyr_bytes, gpitch,
                                   w, h, sw, sh, P->d_xofs, P->d_xa, P->d_yofs, P->d_yb, P->d_px, P->d_defined, P->nwords, P->rhoT, P->k7[0], P->k7[1], P->k7[2], P->k7[3], (const int *)P->d_flags);
            else
                hipLaunchKernelGGL(k_lsd_resize_grad, dim3(gx, (sh + GRAD_ROWS - 1) / GRAD_ROWS, n), dim3(256), 0, st, P->d_blur, w, h, sw, sh, P->d_xofs, P->d_xa, P->d_yofs, P->d_yb,
                                   P->d_px, P->d_defined, P->nwords, P->rho);      // (the split preamble: n <= 16 <= chunk, the blurred images are still there; every frame is formed again)
            { GrowArgs g2 = g; g2.redo = 1; g2.redo_count = P->d_redo; g2.perm = nullptr; hipLaunchKernelGGL(k_lsd_grow, dim3(n), dim3(64), 0, st, g2); }
        }
    } else aw = 0;
    if (aw > 0) {
    } else if (lat) {
        // The LDS request also keeps this one-wave workgroup off the CUs where a frame's five AHC waves sit (k_peac_cluster_heads takes
        // 108 KB): both are bound by instruction issue and a shared SIMD slows both (HVO_LSD_LAT_LDS: bytes requested at least [56 K])
        size_t lds = (size_t)P->nwords * 4, floor_ = 56 * 1024;
        if (P->kn.lat_lds.set) floor_ = (size_t)std::max(P->kn.lat_lds.v, 0);      // HVO_LSD_LAT_LDS
        if (lds < floor_ && floor_ <= 150 * 1024) lds = floor_;
        if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_lsd_grow_lat), lds)) return HVO_ERR_HIP;
        hipLaunchKernelGGL(k_lsd_grow_lat, dim3(n), dim3(64), lds, st, g);
    } else if (P->compact) {
        if (dense) hipLaunchKernelGGL(k_lsd_grow_dense_c, dim3(n), dim3(64), 0, st, g);
        else hipLaunchKernelGGL(k_lsd_grow_c, dim3(n), dim3(64), 0, st, g);
    } else if (dense) hipLaunchKernelGGL(k_lsd_grow_dense, dim3(n), dim3(64), 0, st, g);     // more frames than five waves per SIMD hold
    else hipLaunchKernelGGL(k_lsd_grow, dim3(n), dim3(64), 0, st, g);
    hvo_prof_end(ctx, id);
    if (cull && P->nfeat > CULL_MAXL) return HVO_ERR_UNSUPPORTED;
    for (int c0 = 0; c0 < n; c0 += P->chunk) {
        const int m = std::min(P->chunk, n - c0);
        const uint8_t *gc = gray + (size_t)c0 * O.pyr_bytes;
        const size_t ko = (size_t)c0 * P->nfeat;
        id = hvo_prof_begin(ctx, "lbd_sobel", st);
        if (ctx->readings & HVO_READING_BLUR_FLOAT) {  // GaussianBlur(5 x 5, sigma 1) in the float-kernel reading (readings.hip), then the Sobel pair
            if (!P->d_b5) HVO_HIP(hipMalloc((void **)&P->d_b5, (size_t)P->chunk * w * h));
            if ((rc = readings_gblur_enqueue(st, gc, O.pyr_bytes, gpitch, w, h, P->d_b5, (size_t)w * h, w, m, 5, 1.0, true))) return rc;
            hipLaunchKernelGGL(k_lbd_sobel, dim3((w + 255) / 256, (h + LBD_BLUR_ROWS - 1) / LBD_BLUR_ROWS, m), dim3(256), 0, st, P->d_b5, P->d_dxy, w, h);
        } else
        if (P->d_b5) {                                  // the two-kernel formulation (blurred u8 image materialised), kept for A/B runs (HVO_LBD_SPLIT)
            hipLaunchKernelGGL(k_lbd_blur5, dim3((w + 255) / 256, (h + LBD_BLUR_ROWS - 1) / LBD_BLUR_ROWS, m), dim3(256), 0, st, gc, O.pyr_bytes, gpitch, P->d_b5, w, h, P->k5[0], P->k5[1], P->k5[2]);
            hipLaunchKernelGGL(k_lbd_sobel, dim3((w + 255) / 256, (h + LBD_BLUR_ROWS - 1) / LBD_BLUR_ROWS, m), dim3(256), 0, st, P->d_b5, P->d_dxy, w, h);
        } else
            hipLaunchKernelGGL(k_lbd_blur_sobel, dim3((((w + 3) / 4) * ((h + LBD_BLUR_ROWS - 1) / LBD_BLUR_ROWS) + 255) / 256, 1, m), dim3(256), 0, st, gc, O.pyr_bytes, gpitch, P->d_dxy, w, h,
                               P->k5[0], P->k5[1], P->k5[2]);
        hvo_prof_end(ctx, id);
        id = hvo_prof_begin(ctx, "lbd_desc", st);
        hipLaunchKernelGGL(k_lbd_desc, dim3(P->nfeat, m), dim3(64), 0, st, P->d_dxy, w, h, P->d_kl + ko, P->d_nkl + c0, P->nfeat, P->d_gL, P->d_gG, P->d_desc + ko * 32);
        hvo_prof_end(ctx, id);
        if (cull) {                                     // Frame::cullingLine + the second LBD pass (Frame.cc:934, 952-1116)
            id = hvo_prof_begin(ctx, "lsd_cull", st);
            CullArgs c;
            c.kl = P->d_kl + ko; c.fn = P->d_fn + ko * 3; c.nkl = P->d_nkl + c0; c.tmp = P->d_kl_all + (size_t)c0 * P->maxseg; c.tmp_stride = P->maxseg;
            c.kl_out = P->d_kl2 + ko; c.fn_out = P->d_fn2 + ko * 3; c.nkl_out = P->d_nkl2 + c0;
            c.cap = P->nfeat; c.w = w; c.h = h; c.dis = ctx->cull_dis; c.cos_th = cos(ctx->cull_angle * 0.0174533); c.endpoint_dis = ctx->cull_endpoint;
            const int maxl = (P->nfeat + 63) & ~63;
            if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_cull_lines), cull_lds_bytes(maxl))) return HVO_ERR_HIP;
            hipLaunchKernelGGL(k_cull_lines, dim3(m), dim3(64), cull_lds_bytes(maxl), st, c, maxl);
            hipLaunchKernelGGL(k_lbd_desc, dim3(P->nfeat, m), dim3(64), 0, st, P->d_dxy, w, h, P->d_kl2 + ko, P->d_nkl2 + c0, P->nfeat, P->d_gL, P->d_gG, P->d_desc2 + ko * 32);
            hvo_prof_end(ctx, id);
        }
    }
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

int lsd_download(hvo_ctx *ctx, int n, hvo_frame_out *out, bool culled)
{
    const hipStream_t cs = hvo_copy_stream(ctx, ctx->s_lsd);
    LsdPlan *P = plan_of(ctx);
    if (!P) return HVO_ERR_INVALID_ARG;
    const hvo_keyline *d_kl = culled ? P->d_kl2 : P->d_kl; const uint8_t *d_desc = culled ? P->d_desc2 : P->d_desc;
    const double *d_fn = culled ? P->d_fn2 : P->d_fn; const int *d_nkl = culled ? P->d_nkl2 : P->d_nkl;
    std::vector<int> nk(n), fl(n);
    HVO_HIP(hipMemcpyAsync(nk.data(), d_nkl, n * sizeof(int), hipMemcpyDeviceToHost, cs));
    HVO_HIP(hipMemcpyAsync(fl.data(), P->d_flags, n * sizeof(int), hipMemcpyDeviceToHost, cs));
    HVO_HIP(hipStreamSynchronize(cs));
    std::vector<void *> d0(n, nullptr), d1(n, nullptr), d2(n, nullptr); std::vector<size_t> b0(n, 0), b1(n, 0), b2(n, 0);
    for (int f = 0; f < n; f++) {
        int m = nk[f];
        if (fl[f]) out[f].status = HVO_ERR_CAPACITY;
        if (out[f].kl) {
            if (m > out[f].kl_cap) { m = out[f].kl_cap; out[f].status = HVO_ERR_CAPACITY; }
            if (m > 0) {
                d0[f] = out[f].kl; b0[f] = (size_t)m * sizeof(hvo_keyline);
                if (out[f].ldesc) { d1[f] = out[f].ldesc; b1[f] = (size_t)m * 32; }
                if (out[f].linefn) { d2[f] = out[f].linefn; b2[f] = (size_t)m * 24; }
            }
        }
        out[f].n_kl = m;
    }
    int rc = hvo_staged_d2h(ctx, cs, d_kl, (size_t)P->nfeat * sizeof(hvo_keyline), n, d0.data(), b0.data());
    if (rc) return rc;
    if ((rc = hvo_staged_d2h(ctx, cs, d_desc, (size_t)P->nfeat * 32, n, d1.data(), b1.data()))) return rc;
    if ((rc = hvo_staged_d2h(ctx, cs, d_fn, (size_t)P->nfeat * 24, n, d2.data(), b2.data()))) return rc;
    HVO_HIP(hipStreamSynchronize(cs));
    return HVO_OK;
}

extern "C" int hvo_extract_lsd(hvo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                               hvo_keyline *kl, uint8_t *desc32, double *linefn3, int cap, int *n)
{
    if (!ctx || !n) return HVO_ERR_INVALID_ARG;
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return HVO_OK;           // LineExtractor.cpp:331-332
    if (!kl || !desc32 || !linefn3 || cap < 0 || stride < w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hvo_frame_in in; memset(&in, 0, sizeof(in));
    in.gray = gray; in.gray_stride = stride;
    int rc = orb_upload(ctx, 1, &in, w, h);
    if (rc) return rc;
    ctx->last_stages = 0;                                  // slot 0 of the resident batch has been overwritten
    for (int i = 0; i < ctx->nprof; i++) ctx->prof[i].used = false;
    if ((rc = lsd_run(ctx, 1))) return rc;
    hvo_frame_out out; memset(&out, 0, sizeof(out));
    out.kl = kl; out.ldesc = desc32; out.linefn = linefn3; out.kl_cap = cap;
    if ((rc = lsd_download(ctx, 1, &out))) return rc;
    *n = out.n_kl;
    return out.status;
}

// Frame::ExtractLSD up to and including cullingLine (reference src/Frame.cc:895-934): LINEextractor, then the
// merge of near-collinear segments with the second LBD pass
extern "C" int hvo_extract_lsd_culled(hvo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                                      hvo_keyline *kl, uint8_t *desc32, double *linefn3, int cap, int *n)
{
    if (!ctx || !n) return HVO_ERR_INVALID_ARG;
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return HVO_OK;
    if (!kl || !desc32 || !linefn3 || cap < 0 || stride < w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hvo_frame_in in; memset(&in, 0, sizeof(in));
    in.gray = gray; in.gray_stride = stride;
    int rc = orb_upload(ctx, 1, &in, w, h);
    if (rc) return rc;
    ctx->last_stages = 0;                                  // slot 0 of the resident batch has been overwritten
    for (int i = 0; i < ctx->nprof; i++) ctx->prof[i].used = false;
    if ((rc = lsd_run(ctx, 1, true))) return rc;
    hvo_frame_out out; memset(&out, 0, sizeof(out));
    out.kl = kl; out.ldesc = desc32; out.linefn = linefn3; out.kl_cap = cap;
    if ((rc = lsd_download(ctx, 1, &out, true))) return rc;
    *n = out.n_kl;
    return out.status;
}

extern "C" int hvo_set_line_culling(hvo_ctx *ctx, double dis, double angle_deg, double endpoint_dis)
{
    if (!ctx || !(dis >= 0) || !(angle_deg >= 0) || !(endpoint_dis >= 0)) return HVO_ERR_INVALID_ARG;
    ctx->cull_dis = dis; ctx->cull_angle = angle_deg; ctx->cull_endpoint = endpoint_dis;
    return HVO_OK;
}

// diagnostics (not part of include/hvo.h): per-frame counters of the last k_lsd_grow launch:
// reduce_radius_wave on a list of packed points (y << 16 | x), for the test that compares it with the reference's loop
// (tests/test_lsd_gpu.py::test_reduce_region_radius_closed_form): the list is reordered in place, *n_out = the elements kept,
// released[i] = 1 for every element the radius let go (the reference clears its used flag).
static __global__ __launch_bounds__(64) void k_debug_reduce_radius(int *reg, int n, double xc, double yc, double radSq, int *n_out, unsigned char *released, int w)
{
    __shared__ unsigned long long mask[64]; __shared__ int pre[66];
    const int k = reduce_radius_wave(reg, n, xc, yc, radSq, mask, pre, [&](int a) { released[(a >> 16) * w + (a & 0xFFFF)] = 1; });
    if (threadIdx.x == 0) *n_out = k;
}
extern "C" int hvo_debug_reduce_radius(int *list, int n, double xc, double yc, double rad_sq, int width, int height, unsigned char *released, int *n_kept)
{
    if (!list || n < 1 || n > 4096 || width < 1 || height < 1 || !released || !n_kept) return HVO_ERR_INVALID_ARG;
    int *d = nullptr, *dn = nullptr; unsigned char *dr = nullptr;
    bool ok = hipMalloc((void **)&d, (size_t)n * 4) == hipSuccess && hipMalloc((void **)&dn, 4) == hipSuccess && hipMalloc((void **)&dr, (size_t)width * height) == hipSuccess;
    ok = ok && hipMemcpy(d, list, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemset(dr, 0, (size_t)width * height) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_debug_reduce_radius, dim3(1), dim3(64), 0, 0, d, n, xc, yc, rad_sq, dn, dr, width);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(list, d, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(n_kept, dn, 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(released, dr, (size_t)width * height, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d); (void)hipFree(dn); (void)hipFree(dr);
    return ok ? HVO_OK : HVO_ERR_HIP;
}

// [0] seeds [1] region points grown [2] regions >= min size [3] ticks in region_grow [4] region2rect
// [5] refine [6] whole kernel [7] segments; ticks are 100 MHz wall-clock ticks.
extern "C" int hvo_debug_lsd_stats(hvo_ctx *ctx, int frame, long long *out8)
{
    LsdPlan *P = ctx ? plan_of(ctx) : nullptr;
    if (!P || frame < 0 || frame >= P->batch) return HVO_ERR_INVALID_ARG;
    HVO_HIP(hipMemcpy(out8, P->d_stats + (size_t)frame * 8, 64, hipMemcpyDeviceToHost));
    return HVO_OK;
}

// What the last async line growing (batches of a few frames, the streamed mode) had to fall back on: frames the one-wave kernel grew again
// because the workers' bounded wait expired or no worker sat on the frame's XCD, and workers that found themselves on another XCD than
// their frame's (they count themselves out: exact, but slower -- the round-robin dealing of workgroups b, b + 8, ... to one XCD is an
// assumption about the dispatcher).  Waits for the line stream.  workers_per_frame = 0: the last launch was not async.
extern "C" int hvo_lsd_async_report(hvo_ctx *ctx, int *frames_regrown, int *foreign_workers, int *workers_per_frame)
{
    LsdPlan *P = ctx ? plan_of(ctx) : nullptr;
    if (!ctx) return HVO_ERR_INVALID_ARG;
    if (frames_regrown) *frames_regrown = 0;
    if (foreign_workers) *foreign_workers = 0;
    if (workers_per_frame) *workers_per_frame = 0;
    if (!P || !P->d_actl || P->async_last_n < 1) return HVO_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    HVO_HIP(hipStreamSynchronize(hvo_stream_lsd(ctx)));
    int redo = 0;
    HVO_HIP(hipMemcpy(&redo, P->d_redo, 4, hipMemcpyDeviceToHost));
    std::vector<LaCtl> ctl((size_t)P->async_last_n);
    HVO_HIP(hipMemcpy(ctl.data(), P->d_actl, ctl.size() * sizeof(LaCtl), hipMemcpyDeviceToHost));
    int nf = 0;
    for (const LaCtl &c : ctl) nf += (int)c.n_foreign;
    if (frames_regrown) *frames_regrown = redo;
    if (foreign_workers) *foreign_workers = nf;
    if (workers_per_frame) *workers_per_frame = P->async_last_w;
    return HVO_OK;
}

// diagnostics (not part of include/hvo.h): the control block k_lsd_grow_async left for `frame` (64 words, struct LaCtl): counters and the
// workers' summed ticks in dispatch / growing / waiting for the turn / head work / the tail
extern "C" int hvo_debug_lsd_async(hvo_ctx *ctx, int frame, unsigned *out64)
{
    LsdPlan *P = ctx ? plan_of(ctx) : nullptr;
    if (!P || !P->d_actl || frame < 0 || frame >= P->async_b) return HVO_ERR_INVALID_ARG;
    HVO_HIP(hipMemcpy(out64, (const char *)P->d_actl + (size_t)frame * sizeof(LaCtl), sizeof(LaCtl), hipMemcpyDeviceToHost));
    return HVO_OK;
}

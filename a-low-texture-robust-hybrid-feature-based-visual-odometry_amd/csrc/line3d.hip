// line3d.hip -- Frame::isLineGood (reference src/Frame.cc:1205-1322) for gfx950: the 3-D line of every 2-D key line from the depth
// image (second half).
//   samples along the segment, nearest-pixel depth, back-projection                     src/Frame.cc:1214-1270
//   LINEextractor::compPt3dCov                                                           src/LineExtractor.cpp:44-97
//   LINEextractor::extract3dline_mahdist (+ mah_dist3d_pt_line, verify3dLine, computeLine3d_svd)   src/LineExtractor.cpp:98-327
// One wave per key line.  Lane j owns candidate sample j (<= 21): depth look-up, back-projection, the covariance and its 3x3
// eigen-decomposition, and in every RANSAC / refit round its Mahalanobis distance to the hypothesis; a ballot is the inlier
// set.  The order-dependent parts (partial Fisher-Yates draws, verify3dLine's first-extreme scans, the ordered sums of the
// refit) run on lane 0 over LDS copies of the <= 21 points, in the reference's order.  Same operations in the same order as
// oracle/line3d.c (-ffp-contract=off); the determinism rules (seeded per-line xorshift32 instead of time-seeded rand(),
// Jacobi instead of cv::SVD) are stated there.
#include "hvo_internal.hpp"
#include <string.h>

#define L3_MAXP 24

static __device__ void eig33sym_dev(const double Kin[3][3], double s[3], double V[3][3])
{
    double a[3][3], v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = Kin[i][j];
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off <= 1e-13 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int p = (r == 2) ? 1 : 0, q = (r == 0) ? 1 : 2;
            const double apq = a[p][q];
            if (apq == 0.0) continue;
            const double d = a[q][q] - a[p][p], h = 2.0 * apq;
            const double sg = (d == 0.0 || ((d < 0) == (h < 0))) ? 1.0 : -1.0;
            const double t = sg * fabs(h) / (fabs(d) + sqrt(d * d + h * h));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            const double app = a[p][p], aqq = a[q][q];
            a[p][p] = app - t * apq;
            a[q][q] = aqq + t * apq;
            a[p][q] = a[q][p] = 0.0;
            const int k = 3 - p - q;
            const double akp = a[k][p], akq = a[k][q];
            a[k][p] = a[p][k] = c * akp - sn * akq;
            a[k][q] = a[q][k] = sn * akp + c * akq;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double vip = v[i][p], viq = v[i][q];
                v[i][p] = c * vip - sn * viq;
                v[i][q] = sn * vip + c * viq;
            }
        }
    }
    double d[3] = { a[0][0], a[1][1], a[2][2] };
    int o[3] = { 0, 1, 2 };
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++)
        if (d[o[j]] < d[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
    for (int i = 0; i < 3; i++) { s[i] = d[o[i]]; for (int r = 0; r < 3; r++) V[r][i] = v[r][o[i]]; }
}

static __device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// LINEextractor::mah_dist3d_pt_line (LineExtractor.cpp:186-218): pt = {pos[3], DU[9]}
static __device__ double mah_dist_dev(const double *pos, const double *DU, const double *q1, const double *q2)
{
    const double xa = q1[0], ya = q1[1], za = q1[2], xb = q2[0], yb = q2[1], zb = q2[2];
    const double c1 = DU[0], c2 = DU[1], c3 = DU[2], c4 = DU[3], c5 = DU[4], c6 = DU[5], c7 = DU[6], c8 = DU[7], c9 = DU[8];
    const double x1 = pos[0], x2 = pos[1], x3 = pos[2];
    const double term1 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb)) - (c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term2 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) - (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term3 = ((c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) - (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb))),
                 term4 = (c1 * (x1 - xa) - c1 * (x1 - xb) + c2 * (x2 - ya) - c2 * (x2 - yb) + c3 * (x3 - za) - c3 * (x3 - zb)),
                 term5 = (c4 * (x1 - xa) - c4 * (x1 - xb) + c5 * (x2 - ya) - c5 * (x2 - yb) + c6 * (x3 - za) - c6 * (x3 - zb)),
                 term6 = (c7 * (x1 - xa) - c7 * (x1 - xb) + c8 * (x2 - ya) - c8 * (x2 - yb) + c9 * (x3 - za) - c9 * (x3 - zb));
    return sqrt((term1 * term1 + term2 * term2 + term3 * term3) / (term4 * term4 + term5 * term5 + term6 * term6));
}

// LINEextractor::verify3dLine (LineExtractor.cpp:98-160) over the points whose bit is set in `mask`, in index order (one lane)
static __device__ bool verify_3d_line_dev(const double (*P)[3], unsigned mask, const double *A, const double *B)
{
    int cells = 0;
    double minv = 100, maxv = -100; int i1 = -1, i2 = -1, first = -1;
    const double AB[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
    for (int i = 0; i < L3_MAXP; i++) if ((mask >> i) & 1u) {
        if (first < 0) first = i;
        const double d[3] = { P[i][0] - A[0], P[i][1] - A[1], P[i][2] - A[2] };
        const double v = dot3(d, AB);
        if (v < minv) { minv = v; i1 = i; }
        if (v > maxv) { maxv = v; i2 = i; }
    }
    if (i1 < 0) i1 = first;                                   // idx1 / idx2 start at 0 = the first listed point
    if (i2 < 0) i2 = first;
    const double mid[3] = { (A[0] + B[0]) * 0.5, (A[1] + B[1]) * 0.5, (A[2] + B[2]) * 0.5 };
    double C[3], D[3];
    for (int e = 0; e < 2; e++) {                             // projPt3d2Ln3d (LineExtractor.h:227-235)
        const double *Q = P[e ? i2 : i1];
        const double Bq[3] = { mid[0] + AB[0], mid[1] + AB[1], mid[2] + AB[2] };
        const double ab[3] = { Bq[0] - mid[0], Bq[1] - mid[1], Bq[2] - mid[2] }, ap[3] = { Q[0] - mid[0], Q[1] - mid[1], Q[2] - mid[2] };
        const double t = dot3(ab, ap) / dot3(ab, ab);
        double *o = e ? D : C;
        o[0] = mid[0] + t * ab[0]; o[1] = mid[1] + t * ab[1]; o[2] = mid[2] + t * ab[2];
    }
    const double DC[3] = { D[0] - C[0], D[1] - C[1], D[2] - C[2] };
    const double cd = sqrt(DC[0] * DC[0] + DC[1] * DC[1] + DC[2] * DC[2]);
    if (cd < 0.0000000001) return false;
    for (int i = 0; i < L3_MAXP; i++) if ((mask >> i) & 1u) {
        const double xc[3] = { P[i][0] - C[0], P[i][1] - C[1], P[i][2] - C[2] };
        const double lambda = fabs(dot3(xc, DC) / cd / cd);
        cells |= 1 << (lambda >= 1 ? 9 : (int)(unsigned)floor(lambda * 10));
    }
    double sum = 0;
    for (int i = 0; i < 10; i++) if ((cells >> i) & 1) sum = sum + 1;
    return sum / 10 > 0.7;
}

// LINEextractor::computeLine3d_svd (LineExtractor.cpp:162-184) over the points of `mask`, in index order (one lane)
static __device__ void compute_line3d_dev(const double (*P)[3], unsigned mask, int n, double *mean, double *drct)
{
    mean[0] = mean[1] = mean[2] = 0;
    for (int i = 0; i < L3_MAXP; i++) if ((mask >> i) & 1u) { mean[0] = mean[0] + P[i][0]; mean[1] = mean[1] + P[i][1]; mean[2] = mean[2] + P[i][2]; }
    const double s = 1.0 / n;
    mean[0] = mean[0] * s; mean[1] = mean[1] * s; mean[2] = mean[2] * s;
    double S[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    for (int i = 0; i < L3_MAXP; i++) if ((mask >> i) & 1u) {
        const double q[3] = { P[i][0] - mean[0], P[i][1] - mean[1], P[i][2] - mean[2] };
        for (int a = 0; a < 3; a++) for (int b = a; b < 3; b++) S[a][b] += q[a] * q[b];
    }
    S[1][0] = S[0][1]; S[2][0] = S[0][2]; S[2][1] = S[1][2];
    double w[3], V[3][3];
    eig33sym_dev(S, w, V);
    drct[0] = V[0][2]; drct[1] = V[1][2]; drct[2] = V[2][2];
}

static __device__ __forceinline__ unsigned xs32(unsigned &s) { unsigned x = s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; s = x; return x; }

__global__ __launch_bounds__(64) void k_lines_3d(const hvo_keyline *__restrict__ kl, const int *__restrict__ n_ptr, int n_fixed,
                                                 const uint16_t *__restrict__ depth, int pitch, int w, int h,
                                                 float fx, float fy, float cx, float cy, float dfac, unsigned seed, hvo_line3d *__restrict__ out)
{
    __shared__ double sP[L3_MAXP][3];            // positions of the valid samples, in sample order
    __shared__ double sQ[2][3];                  // the hypothesis (two points of the line)
    __shared__ int s_ctl[4];
    const int li = blockIdx.x, lane = threadIdx.x;
    const int n = n_ptr ? *n_ptr : n_fixed;
    if (li >= n) return;
    hvo_line3d o;
    memset(&o, 0, sizeof(o));
    o.line_eq[0] = o.line_eq[1] = o.line_eq[2] = -1.0f;
    o.line_nor[0] = o.line_nor[1] = o.line_nor[2] = -1.0;
    const float sx = kl[li].sx, sy = kl[li].sy, ex = kl[li].ex, ey = kl[li].ey;
    const float invfx = __fdiv_rn(1.0f, fx), invfy = __fdiv_rn(1.0f, fy);
    const float dxf = __fsub_rn(sx, ex), dyf = __fsub_rn(sy, ey);
    const double len = sqrt((double)dxf * dxf + (double)dyf * dyf);
    const int nsmp = (int)len < 20 ? (int)len : 20;
    const double numSmp = (double)nsmp;
    // ---- the samples: lane j <= nsmp
    bool valid = false; double p[3] = { 0, 0, 0 };
    if (nsmp >= 1 && lane <= nsmp) {
        const double a = 1 - lane / numSmp, b = lane / numSmp;
        const float px = __fadd_rn((float)((double)sx * a), (float)((double)ex * b)), py = __fadd_rn((float)((double)sy * a), (float)((double)ey * b));
        const double ptx = px, pty = py;
        if (!(ptx < 0 || pty < 0 || ptx >= w || pty >= h)) {
            int row, col;
            if (floor(ptx) == ptx && floor(pty) == pty) { col = max((int)(ptx - 1), 0); row = max((int)(pty - 1), 0); }
            else { col = (int)ptx; row = (int)pty; }
            if (!(row < 0 || col < 0 || row >= w || col >= h)) {          // sic (src/Frame.cc:1249): row against cols, col against rows
                const float df = __fmul_rn((float)depth[(size_t)row * pitch + col], dfac);
                if (!((double)df <= 0.01)) {
                    p[2] = df;
                    p[0] = __fsub_rn((float)col, cx) * p[2] * invfx;
                    p[1] = __fsub_rn((float)row, cy) * p[2] * invfy;
                    valid = true;
                }
            }
        }
    }
    const unsigned long long vm = __ballot(valid);
    const int np = __popcll(vm), rank = __popcll(vm & ((1ull << lane) - 1));
    // ---- compPt3dCov of my sample; then point `rank` of the compacted list lives in lane `rank` (and its position in LDS)
    double DUv[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    if (valid) {
        const double f = (double)fx;
        const double zf = p[2] / f, xz = p[0] / p[2], yz = p[1] / p[2];
        const double c1 = 0.00273, c2 = 0.00074, c3 = -0.00058;
        const double sd = c1 * p[2] * p[2] + c2 * p[2] + c3, s2 = sd * sd;
        const double m02 = xz * s2, m12 = yz * s2;
        double K[3][3];
        K[0][0] = zf * zf + m02 * xz; K[0][1] = m02 * yz;            K[0][2] = m02;
        K[1][1] = zf * zf + m12 * yz; K[1][2] = m12;                K[2][2] = s2;
        K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
        double wv[3], U[3][3];
        eig33sym_dev(K, wv, U);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int c = 2 - i;
            const double inv = 1 / sqrt(wv[c]);
            DUv[3 * i + 0] = inv * U[0][c]; DUv[3 * i + 1] = inv * U[1][c]; DUv[3 * i + 2] = inv * U[2][c];
        }
        sP[rank][0] = p[0]; sP[rank][1] = p[1]; sP[rank][2] = p[2];
    }
    __syncthreads();
    // move sample data to lane = compact index: lane i < np takes the point whose rank is i
    double pos[3] = { 0, 0, 0 }, DU[9];
    {
        // source lane of compact index `lane`: the lane-th set bit of vm
        int src = 0;
        { unsigned long long m = vm; for (int k = 0; k < lane && m; k++) m &= m - 1; src = m ? __ffsll((long long)m) - 1 : 0; }
#pragma unroll
        for (int q = 0; q < 9; q++) DU[q] = __shfl(DUv[q], src);
        if (lane < np) { pos[0] = sP[lane][0]; pos[1] = sP[lane][1]; pos[2] = sP[lane][2]; }
    }
    o.n_samples = np;
    unsigned best = 0; int nbest = 0, bestA = 0, bestB = 0;
    double Aout[3] = { 0, 0, 0 }, Bout[3] = { 0, 0, 0 };
    if (np >= 5) {
        const int pairs = (int)(np * (np - 1) * 0.5);
        const int maxIter = pairs < 10 ? pairs : 10;
        int indexes[L3_MAXP];                                  // lane 0's copy is the live one
        for (int i = 0; i < L3_MAXP; i++) indexes[i] = i;
        unsigned rs = seed ^ (0x9E3779B9u * (unsigned)(li + 1)); if (rs == 0) rs = 0x6D2B79F5u;
        for (int it = 0; it < maxIter; it++) {
            // random_unique(indexes, 2) on lane 0 (include/LineExtractor.h:22-36)
            if (lane == 0) {
                int left = np;
                for (int k = 0; k < 2; k++) {
                    const int r = k + (int)((xs32(rs) & 0x7FFFFFFFu) % (unsigned)left);
                    const int t = indexes[k]; indexes[k] = indexes[r]; indexes[r] = t;
                    left--;
                }
                s_ctl[0] = indexes[0]; s_ctl[1] = indexes[1];
            }
            __syncthreads();
            const int ia = s_ctl[0], ib = s_ctl[1];
            const double A[3] = { sP[ia][0], sP[ia][1], sP[ia][2] }, B[3] = { sP[ib][0], sP[ib][1], sP[ib][2] };
            const double ab[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
            __syncthreads();
            if (sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]) < 0.0000000001) continue;
            const bool in = lane < np && mah_dist_dev(pos, DU, A, B) < 3.0;
            const unsigned inl = (unsigned)__ballot(in);
            const int ninl = __popc(inl);
            if (ninl > nbest) {
                if (lane == 0) s_ctl[2] = verify_3d_line_dev(sP, inl, A, B) ? 1 : 0;
                __syncthreads();
                const int okv = s_ctl[2];
                __syncthreads();
                if (okv) { nbest = ninl; best = inl; bestA = ia; bestB = ib; }
            }
            if (nbest > np * 0.6) break;
        }
        if (nbest >= 2) {
            double m[3], d[3];
            for (int c = 0; c < 3; c++) { m[c] = (sP[bestA][c] + sP[bestB][c]) * 0.5; d[c] = sP[bestB][c] - sP[bestA][c]; }
            for (;;) {
                if (lane == 0) {
                    double tm[3], td[3];
                    compute_line3d_dev(sP, best, nbest, tm, td);
                    for (int c = 0; c < 3; c++) { sQ[0][c] = tm[c]; sQ[1][c] = td[c]; }
                }
                __syncthreads();
                const double tm[3] = { sQ[0][0], sQ[0][1], sQ[0][2] }, td[3] = { sQ[1][0], sQ[1][1], sQ[1][2] };
                __syncthreads();
                const double q2[3] = { tm[0] + td[0], tm[1] + td[1], tm[2] + td[2] };
                const bool in = lane < np && mah_dist_dev(pos, DU, tm, q2) < 3.0;
                const unsigned tmp = (unsigned)__ballot(in);
                const int nt = __popc(tmp);
                if (nt > nbest) { nbest = nt; best = tmp; for (int c = 0; c < 3; c++) { m[c] = tm[c]; d[c] = td[c]; } }
                else break;
            }
            // the two end points: first minimum / first maximum of (pos - m) . d over the inliers in index order
            double minv = 100, maxv = -100; int e1 = -1, e2 = -1, first = -1;
            for (int i = 0; i < L3_MAXP; i++) if ((best >> i) & 1u) {
                if (first < 0) first = i;
                const double q[3] = { sP[i][0] - m[0], sP[i][1] - m[1], sP[i][2] - m[2] };
                const double dp = dot3(q, d);
                if (dp < minv) { minv = dp; e1 = i; }
                if (dp > maxv) { maxv = dp; e2 = i; }
            }
            if (e1 < 0) e1 = first;
            if (e2 < 0) e2 = first;
            for (int c = 0; c < 3; c++) { Aout[c] = sP[e1][c]; Bout[c] = sP[e2][c]; }
        }
    }
    if (lane == 0) {
        o.n_inliers = nbest; o.inlier_mask = best;
        const double ab[3] = { Aout[0] - Bout[0], Aout[1] - Bout[1], Aout[2] - Bout[2] };
        if (np >= 5 && sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]) > 0.02) {
            for (int c = 0; c < 3; c++) { o.A[c] = Aout[c]; o.B[c] = Bout[c]; }
            const float le[3] = { (float)(Bout[0] - Aout[0]), (float)(Bout[1] - Aout[1]), (float)(Bout[2] - Aout[2]) };
            const float magn = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(le[0], le[0]), __fmul_rn(le[1], le[1])), __fmul_rn(le[2], le[2])));
            o.line_eq[0] = __fdiv_rn(le[0], magn); o.line_eq[1] = __fdiv_rn(le[1], magn); o.line_eq[2] = __fdiv_rn(le[2], magn);
            o.line_nor[0] = Aout[1] * Bout[2] - Aout[2] * Bout[1];
            o.line_nor[1] = Aout[2] * Bout[0] - Aout[0] * Bout[2];
            o.line_nor[2] = Aout[0] * Bout[1] - Aout[1] * Bout[0];
            o.good = 1;
        }
        out[li] = o;
    }
}

// device-resident form: key lines, their count (d_n, or n_max when null) and the raw depth already in HBM
int lines3d_enqueue(hvo_ctx *ctx, hipStream_t st, const hvo_keyline *d_kl, const int *d_n, int n_max, const uint16_t *d_depth, int pitch, int w, int h,
                    unsigned seed, hvo_line3d *d_out)
{
    if (n_max < 1) return HVO_OK;
    hipLaunchKernelGGL(k_lines_3d, dim3(n_max), dim3(64), 0, st, d_kl, d_n, n_max, d_depth, pitch, w, h, ctx->p.fx, ctx->p.fy, ctx->p.cx, ctx->p.cy,
                       ctx->p.depth_map_factor, seed, d_out);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

// host-array form: a thin wrapper over lines3d_enqueue through the context's staging arena (no allocation per call)
extern "C" int hvo_lines_3d(hvo_ctx *ctx, const hvo_keyline *kl, int n, const uint16_t *depth, int w, int h, int stride, uint32_t seed, hvo_line3d *out)
{
    if (!ctx || n < 0) return HVO_ERR_INVALID_ARG;
    if (n == 0) return HVO_OK;
    if (!kl || !depth || !out || w <= 0 || h <= 0 || stride < 2 * w) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    const size_t b_k = ((size_t)n * sizeof(hvo_keyline) + 255) & ~(size_t)255, b_d = ((size_t)w * h * 2 + 255) & ~(size_t)255;
    char *a = (char *)hvo_call_arena(ctx, b_k + b_d + (size_t)n * sizeof(hvo_line3d));
    if (!a) return HVO_ERR_HIP;
    hvo_keyline *dk = (hvo_keyline *)a; uint16_t *dd = (uint16_t *)(a + b_k); hvo_line3d *dout = (hvo_line3d *)(a + b_k + b_d);
    HVO_HIP(hipMemcpyAsync(dk, kl, (size_t)n * sizeof(hvo_keyline), hipMemcpyHostToDevice, ctx->stream));
    HVO_HIP(hipMemcpy2DAsync(dd, (size_t)w * 2, depth, stride, (size_t)w * 2, h, hipMemcpyHostToDevice, ctx->stream));
    const int rc = lines3d_enqueue(ctx, ctx->stream, dk, nullptr, n, dd, w, w, h, seed, dout);
    if (rc) return rc;
    HVO_HIP(hipMemcpyAsync(out, dout, (size_t)n * sizeof(hvo_line3d), hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    return HVO_OK;
}

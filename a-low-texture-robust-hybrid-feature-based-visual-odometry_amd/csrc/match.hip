// match.hip -- 256-bit Hamming matching on gfx950.
//
//   k_hamming_matrix   ORBmatcher::DescriptorDistance for all pairs   (reference src/ORBmatcher.cc:1676-1692)
//   k_hamming_knn2     cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)      (reference src/LSDmatcher.cpp:811-812, 949)
//
// A descriptor is 4 x u64; distance = sum of popcount(xor).  knn2: one wave per query, each lane
// walks the train set with stride 64 keeping its two best (dist<<16 | idx) keys; a wave-level
// merge then yields the two globally smallest keys -- ascending distance, ties to the lower
// train index, which is what the sequential scan of the reference's matcher produces.
#include "hvo_internal.hpp"
#include <limits.h>
#include <string.h>
#include <vector>

static __device__ __forceinline__ int ham256(const ulonglong4 a, const ulonglong4 b)
{
    return __popcll(a.x ^ b.x) + __popcll(a.y ^ b.y) + __popcll(a.z ^ b.z) + __popcll(a.w ^ b.w);
}

__global__ __launch_bounds__(256) void k_hamming_matrix(const ulonglong4 *__restrict__ q, int nq,
                                                        const ulonglong4 *__restrict__ t, int nt,
                                                        uint16_t *__restrict__ d)
{
    // block = 4 queries x 64 train columns per step
    const int qi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const ulonglong4 a = q[qi];
    for (int j = blockIdx.x * 64 + (threadIdx.x & 63); j < nt; j += gridDim.x * 64)
        d[(size_t)qi * nt + j] = (uint16_t)ham256(a, t[j]);
}

__global__ __launch_bounds__(256) void k_hamming_knn2(const ulonglong4 *__restrict__ q, int nq,
                                                      const ulonglong4 *__restrict__ t, int nt,
                                                      int32_t *__restrict__ idx2, int32_t *__restrict__ dist2)
{
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const ulonglong4 a = q[qi];
    unsigned b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;      // key = dist << 16 | idx (nt <= 65535)
    for (int j = lane; j < nt; j += 64) {
        unsigned k = ((unsigned)ham256(a, t[j]) << 16) | (unsigned)j;
        if (k < b0) { b1 = b0; b0 = k; } else if (k < b1) b1 = k;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned o0 = __shfl_xor(b0, o), o1 = __shfl_xor(b1, o);
        // merge two sorted pairs, keep the two smallest
        unsigned lo = min(b0, o0), hi = max(b0, o0);
        b1 = min(hi, min(b1, o1));
        b0 = lo;
    }
    if (lane == 0) {
        idx2[2 * qi] = b0 == 0xFFFFFFFFu ? -1 : (int)(b0 & 0xFFFF);
        dist2[2 * qi] = b0 == 0xFFFFFFFFu ? INT_MAX : (int)(b0 >> 16);
        idx2[2 * qi + 1] = b1 == 0xFFFFFFFFu ? -1 : (int)(b1 & 0xFFFF);
        dist2[2 * qi + 1] = b1 == 0xFFFFFFFFu ? INT_MAX : (int)(b1 >> 16);
    }
}

static int ensure(hvo_ctx *ctx, void **p, size_t *cap, size_t need)
{
    if (*cap >= need) return HVO_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HVO_HIP(hipMalloc(p, need));
    *cap = need;
    return HVO_OK;
}

static int stage_inputs(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt)
{
    int rc;
    if ((rc = ensure(ctx, (void **)&ctx->d_mq, &ctx->mq_cap, (size_t)nq * 32))) return rc;
    if ((rc = ensure(ctx, (void **)&ctx->d_mt, &ctx->mt_cap, (size_t)nt * 32))) return rc;
    HVO_HIP(hipMemcpyAsync(ctx->d_mq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    HVO_HIP(hipMemcpyAsync(ctx->d_mt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
    return HVO_OK;
}

int match_matrix(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d)
{
    int rc;
    if ((rc = stage_inputs(ctx, q, nq, t, nt))) return rc;
    size_t bytes = (size_t)nq * nt * sizeof(uint16_t);
    if ((rc = ensure(ctx, &ctx->d_mout, &ctx->mout_cap, bytes))) return rc;
    dim3 grd(std::min((nt + 63) / 64, 64), (nq + 3) / 4);
    hipLaunchKernelGGL(k_hamming_matrix, grd, dim3(256), 0, ctx->stream, (const ulonglong4 *)ctx->d_mq, nq,
                       (const ulonglong4 *)ctx->d_mt, nt, (uint16_t *)ctx->d_mout);
    HVO_HIP(hipGetLastError());
    HVO_HIP(hipMemcpyAsync(d, ctx->d_mout, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    return HVO_OK;
}

int match_knn2(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2)
{
    if (nt > 65535) return HVO_ERR_UNSUPPORTED;
    int rc;
    if ((rc = stage_inputs(ctx, q, nq, t, nt))) return rc;
    size_t bytes = (size_t)nq * 2 * sizeof(int32_t);
    if ((rc = ensure(ctx, &ctx->d_mout, &ctx->mout_cap, 2 * bytes))) return rc;
    int32_t *di = (int32_t *)ctx->d_mout, *dd = di + (size_t)nq * 2;
    hipLaunchKernelGGL(k_hamming_knn2, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, (const ulonglong4 *)ctx->d_mq, nq,
                       (const ulonglong4 *)ctx->d_mt, nt, di, dd);
    HVO_HIP(hipGetLastError());
    HVO_HIP(hipMemcpyAsync(idx2, di, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipMemcpyAsync(dist2, dd, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    return HVO_OK;
}

void match_free(hvo_ctx *ctx)
{
    if (ctx->d_mq) (void)hipFree(ctx->d_mq);
    if (ctx->d_mt) (void)hipFree(ctx->d_mt);
    if (ctx->d_mout) (void)hipFree(ctx->d_mout);
    ctx->d_mq = ctx->d_mt = nullptr; ctx->d_mout = nullptr; ctx->mq_cap = ctx->mt_cap = ctx->mout_cap = 0;
}

// =================================================================================================
// Guided search: ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) core
// (reference src/ORBmatcher.cc:1353-1497) with Frame::GetFeaturesInArea (src/Frame.cc:1502-1555) and
// Frame::PosInGrid (src/Frame.cc:1679-1690).
//
// The reference walks the last frame's map points in order; a current-frame feature claimed by an
// earlier point (whose map point has observations) is skipped by later ones.  Best-of-window under
// that dynamic occupancy = the first non-occupied entry of the window's candidates sorted by
// (distance, grid traversal order) -- so the GPU produces, per query, its SBP_K smallest keys
//   key = dist << 32 | (cellX * 48 + cellY) << 16 | index      (cell-major, then insertion order)
// and the O(nq * K) sequential pass over them (plus the 30-bin rotation histogram) is the host
// epilogue in hvo_search_by_projection.  One wave per query.
// =================================================================================================
#define SBP_K 16
#define SBP_LCAP 512
#define SBP_COLS 64
#define SBP_ROWS 48

struct SbpArgs {
    const ulonglong4 *q_desc; const float *q_u, *q_v, *q_radius; const int *q_min_level, *q_max_level; const float *q_ur;
    const hvo_keypoint *t_kp; const float *t_uright; const uint8_t *t_occ; const ulonglong4 *t_desc;
    int nq, nt; float mnMinX, mnMinY, invW, invH;
    unsigned long long *out_key; int *out_cnt;
};

__global__ __launch_bounds__(256) void k_search_by_projection(SbpArgs a)
{
    __shared__ unsigned long long list[4][SBP_LCAP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wv;
    if (qi >= a.nq) return;
    unsigned long long *L = list[wv];
    const float x = a.q_u[qi], y = a.q_v[qi], r = a.q_radius[qi];
    const int minLevel = a.q_min_level[qi], maxLevel = a.q_max_level[qi];
    // GetFeaturesInArea cell range (Frame.cc:1507-1521)
    int nMinCellX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.mnMinX), r), a.invW)));
    int nMaxCellX = min(SBP_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.mnMinX), r), a.invW)));
    int nMinCellY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.mnMinY), r), a.invH)));
    int nMaxCellY = min(SBP_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.mnMinY), r), a.invH)));
    const bool empty = nMinCellX >= SBP_COLS || nMaxCellX < 0 || nMinCellY >= SBP_ROWS || nMaxCellY < 0;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const ulonglong4 qd = a.q_desc[qi];
    const float qur = a.q_ur ? a.q_ur[qi] : -1.f;
    int n = 0;
    for (int base = 0; base < a.nt && !empty; base += 64) {
        const int j = base + lane;
        bool ok = false; unsigned long long key = 0;
        if (j < a.nt) {
            const hvo_keypoint kp = a.t_kp[j];
            // PosInGrid (Frame.cc:1681-1682): round half away from zero
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, a.mnMinX), a.invW)), py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, a.mnMinY), a.invH));
            ok = px >= nMinCellX && px <= nMaxCellX && py >= nMinCellY && py <= nMaxCellY;      // implies it is inside the grid
            if (ok && bCheckLevels) ok = !(kp.octave < minLevel) && !(maxLevel >= 0 && kp.octave > maxLevel);
            if (ok) ok = fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r;
            if (ok && a.t_occ) ok = a.t_occ[j] == 0;
            if (ok && a.t_uright && a.q_ur) { const float ur2 = a.t_uright[j]; if (ur2 > 0) ok = !(fabsf(__fsub_rn(qur, ur2)) > r); }
            if (ok) {
                const ulonglong4 td = a.t_desc[j];
                const unsigned d = __popcll(qd.x ^ td.x) + __popcll(qd.y ^ td.y) + __popcll(qd.z ^ td.z) + __popcll(qd.w ^ td.w);
                key = ((unsigned long long)d << 32) | ((unsigned long long)(px * SBP_ROWS + py) << 16) | (unsigned long long)j;
            }
        }
        const unsigned long long m = __ballot(ok);
        if (ok) { const int p = n + __popcll(m & ((1ull << lane) - 1)); if (p < SBP_LCAP) L[p] = key; }
        n += __popcll(m);
    }
    __syncthreads();
    const int nl = min(n, SBP_LCAP);
    // K rounds of wave-min extraction
    for (int k = 0; k < SBP_K; k++) {
        unsigned long long best = ~0ull; int bi = -1;
        for (int i = lane; i < nl; i += 64) { const unsigned long long v = L[i]; if (v < best) { best = v; bi = i; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ob < best) { best = ob; bi = oi; }
        }
        if (lane == 0) { a.out_key[(size_t)qi * SBP_K + k] = best; if (bi >= 0) L[bi] = ~0ull; }
        __syncthreads();
    }
    if (lane == 0) a.out_cnt[qi] = n;
}

__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const hvo_keypoint *__restrict__ kp, const hvo_keypoint *__restrict__ kpun, int n,
                                                          const uint16_t *__restrict__ depth, int pitch, int w, int h, float dfac, float bf,
                                                          float *__restrict__ uright, float *__restrict__ zdepth)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float ur = -1.f, z = -1.f;
    const int v = (int)kp[i].y, u = (int)kp[i].x;            // imDepth.at<float>(v, u) with float v,u (Frame.cc:1950-1953)
    if (u >= 0 && v >= 0 && u < w && v < h) {
        const float d = __fmul_rn((float)depth[(size_t)v * pitch + u], dfac);
        if (d > 0 && (double)d < 7.0) { z = d; ur = __fsub_rn(kpun[i].x, __fdiv_rn(bf, d)); }
    }
    uright[i] = ur; zdepth[i] = z;
}

template <class T> static int up(hvo_ctx *ctx, T **d, const T *h, size_t n)
{
    *d = nullptr;
    if (!h || !n) return HVO_OK;
    HVO_HIP(hipMalloc((void **)d, n * sizeof(T)));
    HVO_HIP(hipMemcpyAsync(*d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return HVO_OK;
}

int match_search_by_projection(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                               const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur,
                               const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                               float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, unsigned long long *keys, int *cnt)
{
    SbpArgs a; memset(&a, 0, sizeof(a));
    uint8_t *dq = nullptr, *dt = nullptr, *docc = nullptr; float *du = nullptr, *dv = nullptr, *dr = nullptr, *dur = nullptr, *dtu = nullptr;
    int *dmin = nullptr, *dmax = nullptr; hvo_keypoint *dkp = nullptr; unsigned long long *dkeys = nullptr; int *dcnt = nullptr;
    int rc = HVO_OK;
    void *all[16]; int na = 0;
#define UP(dst, src, n) do { if ((rc = up(ctx, &dst, src, n))) goto done; all[na++] = dst; } while (0)
    UP(dq, q_desc, (size_t)nq * 32); UP(dt, t_desc, (size_t)nt * 32); UP(du, q_u, (size_t)nq); UP(dv, q_v, (size_t)nq); UP(dr, q_radius, (size_t)nq);
    UP(dmin, q_min_level, (size_t)nq); UP(dmax, q_max_level, (size_t)nq); UP(dur, q_ur, (size_t)nq); UP(dkp, t_kp, (size_t)nt);
    UP(dtu, t_uright, (size_t)nt); UP(docc, t_occupied, (size_t)nt);
#undef UP
    if (hipMalloc((void **)&dkeys, (size_t)nq * SBP_K * 8) != hipSuccess || hipMalloc((void **)&dcnt, (size_t)nq * 4) != hipSuccess) { rc = HVO_ERR_HIP; goto done; }
    a.q_desc = (const ulonglong4 *)dq; a.q_u = du; a.q_v = dv; a.q_radius = dr; a.q_min_level = dmin; a.q_max_level = dmax; a.q_ur = dur;
    a.t_kp = dkp; a.t_uright = dtu; a.t_occ = docc; a.t_desc = (const ulonglong4 *)dt; a.nq = nq; a.nt = nt;
    a.mnMinX = mnMinX; a.mnMinY = mnMinY;
    a.invW = (float)SBP_COLS / (mnMaxX - mnMinX); a.invH = (float)SBP_ROWS / (mnMaxY - mnMinY);     // Frame.cc:184-185
    a.out_key = dkeys; a.out_cnt = dcnt;
    hipLaunchKernelGGL(k_search_by_projection, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, a);
    if (hipMemcpyAsync(keys, dkeys, (size_t)nq * SBP_K * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(cnt, dcnt, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) rc = HVO_ERR_HIP;
done:
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < na; i++) if (all[i]) (void)hipFree(all[i]);
    if (dkeys) (void)hipFree(dkeys);
    if (dcnt) (void)hipFree(dcnt);
    return rc;
}

int match_stereo_from_rgbd(hvo_ctx *ctx, const hvo_keypoint *kp, const hvo_keypoint *kpun, int n, const uint16_t *depth, int w, int h, int stride,
                           float bf, float *uright, float *zdepth)
{
    hvo_keypoint *dk = nullptr, *dku = nullptr; uint16_t *dd = nullptr; float *dur = nullptr, *dz = nullptr;
    int rc = HVO_OK;
    if (hipMalloc((void **)&dk, (size_t)n * sizeof(hvo_keypoint)) != hipSuccess || hipMalloc((void **)&dku, (size_t)n * sizeof(hvo_keypoint)) != hipSuccess ||
        hipMalloc((void **)&dd, (size_t)w * h * 2) != hipSuccess || hipMalloc((void **)&dur, (size_t)n * 4) != hipSuccess || hipMalloc((void **)&dz, (size_t)n * 4) != hipSuccess) rc = HVO_ERR_HIP;
    if (!rc) {
        (void)hipMemcpyAsync(dk, kp, (size_t)n * sizeof(hvo_keypoint), hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(dku, kpun, (size_t)n * sizeof(hvo_keypoint), hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpy2DAsync(dd, (size_t)w * 2, depth, stride, (size_t)w * 2, h, hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dk, dku, n, dd, w, w, h, ctx->p.depth_map_factor, bf, dur, dz);
        (void)hipMemcpyAsync(uright, dur, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(zdepth, dz, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = HVO_ERR_HIP;
    }
    void *ptrs[] = { dk, dku, dd, dur, dz };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    return rc;
}

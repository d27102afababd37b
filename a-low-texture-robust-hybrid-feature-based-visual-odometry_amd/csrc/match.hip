// match.hip -- 256-bit Hamming matching on gfx950.
//
//   k_hamming_matrix        ORBmatcher::DescriptorDistance for all pairs   (reference src/ORBmatcher.cc:1676-1692)
//   k_hamming_knn2          cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)      (reference src/LSDmatcher.cpp:811-812, 949)
//   k_search_by_projection  ranked window candidates of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:1353-1497, 45-132)
//   k_sbp_epilogue          the reference's sequential pass over the queries (occupancy, ratio test, 30-bin rotation
//                           histogram + ComputeThreeMaxima, src/ORBmatcher.cc:1425-1497, 1630-1673), one wave
//   k_frame_bf_epilogue     LSDmatcher::FrameBFMatch's tests on a knn-2 table with lineDescriptorMAD's threshold
//                           (src/LSDmatcher.cpp:942-966, 1110-1135); k_mutual_check = SearchDouble's two-way check (902-939)
//
// A descriptor is 4 x u64; distance = sum of popcount(xor).  knn2: one wave per query, each lane
// walks the train set with stride 64 keeping its two best (dist<<16 | idx) keys; a wave-level
// merge then yields the two globally smallest keys -- ascending distance, ties to the lower
// train index, which is what the sequential scan of the reference's matcher produces.
//
// Every pipeline exists in a device-resident form (match_*_enqueue: device pointers in, device results out, no
// allocation, no synchronisation) used by the streamed-sequence mode (stream.hip), and behind the host-array entry
// points of include/hvo.h, which stage their arguments through a grow-only arena (one device + one pinned host block
// per context) instead of a hipMalloc / hipFree pair per argument.
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

// ------------------------------------------------------------------------------------------------
// staging arena: one device block and one pinned host block per context, grown on demand
// ------------------------------------------------------------------------------------------------
struct MatchArena {
    char *d = nullptr, *h = nullptr; size_t dcap = 0, hcap = 0, doff = 0, hoff = 0;
};
static MatchArena *arena_of(hvo_ctx *ctx)
{
    if (!ctx->marena) ctx->marena = new MatchArena();
    return (MatchArena *)ctx->marena;
}
// makes room for `dbytes` of device and `hbytes` of pinned host staging and rewinds both
static int arena_begin(hvo_ctx *ctx, size_t dbytes, size_t hbytes)
{
    MatchArena *A = arena_of(ctx);
    dbytes += 4096; hbytes += 4096;                            // alignment slack
    if (A->dcap < dbytes) {
        HVO_HIP(hipStreamSynchronize(ctx->stream));
        if (A->d) (void)hipFree(A->d);
        A->d = nullptr; A->dcap = 0;
        const size_t want = dbytes + dbytes / 2;
        HVO_HIP(hipMalloc((void **)&A->d, want));
        A->dcap = want;
    }
    if (A->hcap < hbytes) {
        HVO_HIP(hipStreamSynchronize(ctx->stream));
        if (A->h) (void)hipHostFree(A->h);
        A->h = nullptr; A->hcap = 0;
        const size_t want = hbytes + hbytes / 2;
        HVO_HIP(hipHostMalloc((void **)&A->h, want, hipHostMallocDefault));
        A->hcap = want;
    }
    A->doff = A->hoff = 0;
    return HVO_OK;
}
template <class T> static T *arena_dev(hvo_ctx *ctx, size_t n)
{
    MatchArena *A = arena_of(ctx);
    A->doff = (A->doff + 63) & ~(size_t)63;
    T *p = (T *)(A->d + A->doff);
    A->doff += n * sizeof(T);
    return p;
}
template <class T> static T *arena_host(hvo_ctx *ctx, size_t n)
{
    MatchArena *A = arena_of(ctx);
    A->hoff = (A->hoff + 63) & ~(size_t)63;
    T *p = (T *)(A->h + A->hoff);
    A->hoff += n * sizeof(T);
    return p;
}
// host array -> pinned staging -> device (async on the ctx stream); returns the device pointer (nullptr for a null source)
template <class T> static T *arena_up(hvo_ctx *ctx, const T *src, size_t n)
{
    if (!src || !n) return nullptr;
    T *h = arena_host<T>(ctx, n), *d = arena_dev<T>(ctx, n);
    memcpy(h, src, n * sizeof(T));
    (void)hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
    return d;
}
#define AL(n, T) (((size_t)(n) * sizeof(T) + 127) & ~(size_t)63)

int match_matrix(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *d)
{
    const size_t obytes = (size_t)nq * nt * sizeof(uint16_t);
    int rc = arena_begin(ctx, AL(nq * 32, char) + AL(nt * 32, char) + AL(obytes, char), AL(nq * 32, char) + AL(nt * 32, char));
    if (rc) return rc;
    const uint8_t *dq = arena_up(ctx, q, (size_t)nq * 32), *dt = arena_up(ctx, t, (size_t)nt * 32);
    uint16_t *dd = arena_dev<uint16_t>(ctx, (size_t)nq * nt);
    dim3 grd(std::min((nt + 63) / 64, 64), (nq + 3) / 4);
    hipLaunchKernelGGL(k_hamming_matrix, grd, dim3(256), 0, ctx->stream, (const ulonglong4 *)dq, nq, (const ulonglong4 *)dt, nt, dd);
    HVO_HIP(hipGetLastError());
    HVO_HIP(hipMemcpyAsync(d, dd, obytes, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    return HVO_OK;
}

int match_knn2_enqueue(hipStream_t st, const uint8_t *dq, int nq, const uint8_t *dt, int nt, int32_t *d_idx2, int32_t *d_dist2)
{
    if (nq < 1) return HVO_OK;
    hipLaunchKernelGGL(k_hamming_knn2, dim3((nq + 3) / 4), dim3(256), 0, st, (const ulonglong4 *)dq, nq, (const ulonglong4 *)dt, nt, d_idx2, d_dist2);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

int match_knn2(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx2, int32_t *dist2)
{
    if (nt > 65535) return HVO_ERR_UNSUPPORTED;
    const size_t ob = (size_t)nq * 2 * sizeof(int32_t);
    int rc = arena_begin(ctx, AL(nq * 32, char) + AL(nt * 32, char) + 2 * AL(ob, char), AL(nq * 32, char) + AL(nt * 32, char) + 2 * AL(ob, char));
    if (rc) return rc;
    const uint8_t *dq = arena_up(ctx, q, (size_t)nq * 32), *dt = arena_up(ctx, t, (size_t)nt * 32);
    int32_t *di = arena_dev<int32_t>(ctx, (size_t)nq * 2), *dd = arena_dev<int32_t>(ctx, (size_t)nq * 2);
    int32_t *hi = arena_host<int32_t>(ctx, (size_t)nq * 2), *hd = arena_host<int32_t>(ctx, (size_t)nq * 2);
    if ((rc = match_knn2_enqueue(ctx->stream, dq, nq, dt, nt, di, dd))) { ctx->last_error = "k_hamming_knn2 launch"; return rc; }
    HVO_HIP(hipMemcpyAsync(hi, di, ob, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipMemcpyAsync(hd, dd, ob, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(idx2, hi, ob); memcpy(dist2, hd, ob);
    return HVO_OK;
}

void match_free(hvo_ctx *ctx)
{
    MatchArena *A = (MatchArena *)ctx->marena;
    if (A) {
        if (A->d) (void)hipFree(A->d);
        if (A->h) (void)hipHostFree(A->h);
        delete A;
        ctx->marena = nullptr;
    }
}

// =================================================================================================
// LSDmatcher::FrameBFMatch on a knn-2 table (reference src/LSDmatcher.cpp:942-966) with lineDescriptorMAD's
// nn12 threshold (1110-1135).  The two medians are order statistics (the reference sorts, the sorts' tie order
// cannot change them): element k = n/2 of the descending (first) and of the ascending (second) order, found by
// counting -- x is the k-th of the descending order iff  #(v > x) <= k < #(v >= x).
// One workgroup; v lives in a global scratch of 2*n floats (n is a few hundred lines).
// =================================================================================================
__global__ __launch_bounds__(256) void k_frame_bf_epilogue(const int32_t *__restrict__ idx2, const int32_t *__restrict__ dist2, int n1, int n2,
                                                           float TH, float nnratio, float *__restrict__ v, int32_t *__restrict__ m12, int *__restrict__ nmatch)
{
    __shared__ float s_med[2];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, k = n1 / 2;
    if (tid == 0) s_cnt = 0;
    if (n2 < 2) {                                              // knnMatch(k = 2) needs two train descriptors
        for (int i = tid; i < n1; i += 256) m12[i] = -1;
        if (tid == 0) *nmatch = 0;
        return;
    }
    float *v1 = v, *v2 = v + n1;
    for (int i = tid; i < n1; i += 256) v1[i] = __fsub_rn((float)dist2[2 * i + 1], (float)dist2[2 * i]);
    __syncthreads();
    for (int i = tid; i < n1; i += 256) {
        const float x = v1[i];
        int gt = 0, ge = 0;
        for (int q = 0; q < n1; q++) { const float y = v1[q]; gt += y > x; ge += y >= x; }
        if (gt <= k && k < ge) s_med[0] = x;                   // every thread that qualifies holds the same value
    }
    __syncthreads();
    const double nn12_median = (double)s_med[0];
    for (int i = tid; i < n1; i += 256) v2[i] = fabsf((float)((double)v1[i] - nn12_median));
    __syncthreads();
    for (int i = tid; i < n1; i += 256) {
        const float x = v2[i];
        int lt = 0, le = 0;
        for (int q = 0; q < n1; q++) { const float y = v2[q]; lt += y < x; le += y <= x; }
        if (lt <= k && k < le) s_med[1] = x;
    }
    __syncthreads();
    double nn12_th = 1.4826 * (double)s_med[1];
    nn12_th = nn12_th * 0.5;
    int mine = 0;
    for (int i = tid; i < n1; i += 256) {
        const float d0 = (float)dist2[2 * i], d1 = (float)dist2[2 * i + 1];
        const double dist_12 = (double)__fsub_rn(d1, d0);
        const bool ok = dist_12 > nn12_th && d0 < TH && d0 < __fmul_rn(nnratio, d1);
        m12[i] = ok ? idx2[2 * i] : -1;
        mine += ok;
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) *nmatch = s_cnt;
}

// LSDmatcher::SearchDouble's mutual check (reference src/LSDmatcher.cpp:920-936): i -> j survives only if j -> i
__global__ __launch_bounds__(256) void k_mutual_check(int32_t *__restrict__ m12, const int32_t *__restrict__ m21, int n1, int *__restrict__ nmatch)
{
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n1; i += 256) {
        const int j = m12[i];
        if (j >= 0) { if (m21[j] != i) m12[i] = -1; else mine++; }
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) *nmatch = s_cnt;
}

// LSDmatcher::matchNNR's ratio test on a knn-2 table (reference src/LSDmatcher.cpp:815-823)
__global__ __launch_bounds__(256) void k_nnr_epilogue(const int32_t *__restrict__ idx2, const int32_t *__restrict__ dist2, int n1, int n2, float nnr,
                                                      int32_t *__restrict__ m12, int *__restrict__ nmatch)
{
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n1; i += 256) {
        // the reference indexes matches_[idx][1] unconditionally: needs n2 >= 2
        const bool ok = n2 >= 2 && (float)dist2[2 * i] < __fmul_rn((float)dist2[2 * i + 1], nnr);
        m12[i] = ok ? idx2[2 * i] : -1;
        mine += ok;
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) *nmatch = s_cnt;
}

// device-resident line matching.  mode 0: matchNNR (th unused, nnratio = nnr); 1: FrameBFMatch; 2: SearchDouble (both
// directions + mutual check).  scratch: see match_lines_scratch_bytes.
size_t match_lines_scratch_bytes(int n1, int n2)
{
    const size_t n = (size_t)std::max(n1, n2) + 16;
    return 4 * AL(2 * n, int32_t) + AL(2 * n, float) + AL(n, int32_t) + 256;
}
int match_lines_enqueue(hipStream_t st, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float TH, float nnratio, int mode,
                        void *scratch, int32_t *d_m12, int *d_nmatch)
{
    const int mutual = mode == 2;
    if (n1 < 1) return HVO_OK;
    char *s = (char *)scratch;
    const size_t n = (size_t)std::max(n1, n2) + 16;
    int32_t *idx = (int32_t *)s; s += AL(2 * n, int32_t);
    int32_t *dist = (int32_t *)s; s += AL(2 * n, int32_t);
    int32_t *idx_b = (int32_t *)s; s += AL(2 * n, int32_t);
    int32_t *dist_b = (int32_t *)s; s += AL(2 * n, int32_t);
    float *v = (float *)s; s += AL(2 * n, float);
    int32_t *m21 = (int32_t *)s; s += AL(n, int32_t);
    int *nm21 = (int *)s;
    if (n2 >= 1) hipLaunchKernelGGL(k_hamming_knn2, dim3((n1 + 3) / 4), dim3(256), 0, st, (const ulonglong4 *)d1, n1, (const ulonglong4 *)d2, n2, idx, dist);
    if (mode == 0) hipLaunchKernelGGL(k_nnr_epilogue, dim3(1), dim3(256), 0, st, idx, dist, n1, n2, nnratio, d_m12, d_nmatch);
    else hipLaunchKernelGGL(k_frame_bf_epilogue, dim3(1), dim3(256), 0, st, idx, dist, n1, n2, TH, nnratio, v, d_m12, d_nmatch);
    if (mutual && n2 >= 1) {
        hipLaunchKernelGGL(k_hamming_knn2, dim3((n2 + 3) / 4), dim3(256), 0, st, (const ulonglong4 *)d2, n2, (const ulonglong4 *)d1, n1, idx_b, dist_b);
        hipLaunchKernelGGL(k_frame_bf_epilogue, dim3(1), dim3(256), 0, st, idx_b, dist_b, n2, n1, TH, nnratio, v, m21, nm21);
        hipLaunchKernelGGL(k_mutual_check, dim3(1), dim3(256), 0, st, d_m12, m21, n1, d_nmatch);
    }
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

int match_lines(hvo_ctx *ctx, const uint8_t *d1, int n1, const uint8_t *d2, int n2, float th, float nnratio, int mode, int32_t *m12, int *n_matches)
{
    if (n1 > 65535 || n2 > 65535) return HVO_ERR_UNSUPPORTED;
    const size_t sb = match_lines_scratch_bytes(n1, n2);
    int rc = arena_begin(ctx, AL(n1 * 32, char) + AL(n2 * 32, char) + sb + AL(n1 + 1, int32_t) + 64, AL(n1 * 32, char) + AL(n2 * 32, char) + AL(n1 + 1, int32_t) + 64);
    if (rc) return rc;
    const uint8_t *a = arena_up(ctx, d1, (size_t)n1 * 32), *b = arena_up(ctx, d2, (size_t)n2 * 32);
    void *scratch = arena_dev<char>(ctx, sb);
    int32_t *dm = arena_dev<int32_t>(ctx, (size_t)n1 + 1); int *dn = dm + n1;
    int32_t *hm = arena_host<int32_t>(ctx, (size_t)n1 + 1);
    if ((rc = match_lines_enqueue(ctx->stream, a, n1, b, n2, th, nnratio, mode, scratch, dm, dn))) { ctx->last_error = "line matching launch"; return rc; }
    HVO_HIP(hipMemcpyAsync(hm, dm, ((size_t)n1 + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(m12, hm, (size_t)n1 * sizeof(int32_t));
    *n_matches = hm[n1];
    return HVO_OK;
}

// =================================================================================================
// Guided search: ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) core
// (reference src/ORBmatcher.cc:1353-1497) and SearchByProjection(F, vpMapPoints, th) core (45-132) with
// Frame::GetFeaturesInArea (src/Frame.cc:1502-1555) and Frame::PosInGrid (src/Frame.cc:1679-1690).
//
// The reference walks the queries in order; a current-frame feature claimed by an earlier query (whose map point
// has observations) is skipped by later ones.  Best-of-window under that dynamic occupancy = the first
// non-occupied entry of the window's candidates sorted by (distance, grid traversal order) -- so
// k_search_by_projection produces, per query and in parallel (one wave per query), its SBP_K smallest keys
//   key = dist << 32 | (cellX * 48 + cellY) << 16 | index      (cell-major, then insertion order)
// and k_sbp_epilogue -- ONE wave -- walks the queries in order over those keys with the occupancy bit set in LDS:
// lane k tests candidate k, a ballot picks the first (and, for the local-map variant, the second) free one.  A query
// whose ranked candidates are all claimed although its window holds more is searched again on the spot (the whole wave
// scans the train features against the occupancy reached so far; exact, rare).  The 30-bin rotation histogram,
// ComputeThreeMaxima and the cull of the other bins run in the same kernel.
// =================================================================================================
#define SBP_K HVO_SBP_K
#define SBP_LCAP 512
#define SBP_COLS 64
#define SBP_ROWS 48
#define SBP_MAXQ 16384            // queries per call (rotation bins are kept in LDS)

struct SbpQuery { float x, y, r, qur; int minLevel, maxLevel, cx0, cx1, cy0, cy1; bool empty, checkLevels; ulonglong4 qd; };

static __device__ __forceinline__ SbpQuery sbp_query(const SbpDev &a, int qi)
{
    SbpQuery q;
    q.x = a.q_u[qi]; q.y = a.q_v[qi]; q.r = a.q_radius[qi];
    q.minLevel = a.q_min_level[qi]; q.maxLevel = a.q_max_level[qi];
    const float invW = (float)SBP_COLS / (a.mnMaxX - a.mnMinX), invH = (float)SBP_ROWS / (a.mnMaxY - a.mnMinY);     // Frame.cc:184-185
    // GetFeaturesInArea cell range (Frame.cc:1507-1521)
    q.cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.x, a.mnMinX), q.r), invW)));
    q.cx1 = min(SBP_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.x, a.mnMinX), q.r), invW)));
    q.cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.y, a.mnMinY), q.r), invH)));
    q.cy1 = min(SBP_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.y, a.mnMinY), q.r), invH)));
    q.empty = q.cx0 >= SBP_COLS || q.cx1 < 0 || q.cy0 >= SBP_ROWS || q.cy1 < 0;
    q.checkLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
    const uint8_t *dsrc = a.q_desc + 32 * (size_t)(a.q_desc_index ? a.q_desc_index[qi] : qi);
    q.qd = *reinterpret_cast<const ulonglong4 *>(dsrc);
    q.qur = a.q_ur ? a.q_ur[qi] : -1.f;
    return q;
}

// is train feature j a candidate of query q (all of GetFeaturesInArea's and SearchByProjection's tests but the occupancy)?
static __device__ __forceinline__ bool sbp_candidate(const SbpDev &a, const SbpQuery &q, int j, unsigned long long &key)
{
    const float invW = (float)SBP_COLS / (a.mnMaxX - a.mnMinX), invH = (float)SBP_ROWS / (a.mnMaxY - a.mnMinY);
    const hvo_keypoint kp = a.t_kp[j];
    // PosInGrid (Frame.cc:1681-1682): round half away from zero
    const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, a.mnMinX), invW)), py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, a.mnMinY), invH));
    bool ok = px >= q.cx0 && px <= q.cx1 && py >= q.cy0 && py <= q.cy1;      // implies it is inside the grid
    if (ok && q.checkLevels) ok = !(kp.octave < q.minLevel) && !(q.maxLevel >= 0 && kp.octave > q.maxLevel);
    if (ok) ok = fabsf(__fsub_rn(kp.x, q.x)) < q.r && fabsf(__fsub_rn(kp.y, q.y)) < q.r;
    if (ok && a.t_uright && a.q_ur) { const float ur2 = a.t_uright[j]; if (ur2 > 0) ok = !(fabsf(__fsub_rn(q.qur, ur2)) > q.r); }
    if (ok) {
        const ulonglong4 td = reinterpret_cast<const ulonglong4 *>(a.t_desc)[j];
        const unsigned d = (unsigned)ham256(q.qd, td);
        key = ((unsigned long long)d << 32) | ((unsigned long long)(px * SBP_ROWS + py) << 16) | (unsigned long long)j;
    }
    return ok;
}

// the SBP_K smallest keys of L[0 .. nl) -> out[0 .. SBP_K) (~0 padded), by SBP_K rounds of wave-min extraction
static __device__ __forceinline__ void sbp_topk(unsigned long long *L, int nl, unsigned long long *out, int lane)
{
    for (int k = 0; k < SBP_K; k++) {
        unsigned long long best = ~0ull; int bi = -1;
        for (int i = lane; i < nl; i += 64) { const unsigned long long v = L[i]; if (v < best) { best = v; bi = i; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ob < best) { best = ob; bi = oi; }
        }
        if (lane == 0) { out[k] = best; if (bi >= 0) L[bi] = ~0ull; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_search_by_projection(SbpDev a)
{
    __shared__ unsigned long long L[SBP_LCAP];
    __shared__ unsigned long long topk[SBP_K];
    const int lane = threadIdx.x, qi = blockIdx.x;             // one wave = one workgroup = one query
    const SbpQuery q = sbp_query(a, qi);
    int n = 0, nl = 0;
    const int nt = q.empty ? 0 : a.nt;
    for (int base = 0; base < nt; base += 64) {
        // a window with more candidates than the list holds keeps its running top-K: the ranks are exact for any count
        if (nl + 64 > SBP_LCAP) {
            __syncthreads();
            sbp_topk(L, nl, topk, lane);
            if (lane < SBP_K) L[lane] = topk[lane];
            nl = SBP_K;
            __syncthreads();
        }
        const int j = base + lane;
        bool ok = false; unsigned long long key = 0;
        if (j < nt) {
            ok = sbp_candidate(a, q, j, key);
            if (ok && a.t_occ) ok = a.t_occ[j] == 0;
        }
        const unsigned long long m = __ballot(ok);
        if (ok) L[nl + __popcll(m & ((1ull << lane) - 1))] = key;
        nl += __popcll(m); n += __popcll(m);
    }
    __syncthreads();
    sbp_topk(L, nl, topk, lane);
    if (lane < SBP_K) a.keys[(size_t)qi * SBP_K + lane] = topk[lane];
    if (lane == 0) a.cnt[qi] = n;
}

// best and second best free candidates of query qi under the occupancy reached so far, over ALL train features
static __device__ void sbp_rescan(const SbpDev &a, int qi, const unsigned *occ, int lane, unsigned long long &k1, unsigned long long &k2)
{
    const SbpQuery q = sbp_query(a, qi);
    unsigned long long b0 = ~0ull, b1 = ~0ull;
    if (!q.empty) for (int j = lane; j < a.nt; j += 64) {
        unsigned long long key = 0;
        if (sbp_candidate(a, q, j, key) && !((occ[j >> 5] >> (j & 31)) & 1u)) { if (key < b0) { b1 = b0; b0 = key; } else if (key < b1) b1 = key; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long o0 = __shfl_xor(b0, o), o1 = __shfl_xor(b1, o);
        const unsigned long long lo = b0 < o0 ? b0 : o0, hi = b0 < o0 ? o0 : b0;
        const unsigned long long m1 = b1 < o1 ? b1 : o1;
        b1 = hi < m1 ? hi : m1; b0 = lo;
    }
    k1 = b0; k2 = b1;
}

__global__ __launch_bounds__(64) void k_sbp_epilogue(SbpDev a)
{
    __shared__ unsigned occ[2048];                // bit j: feature j holds an observed map point (t_occ) or was claimed by an earlier query
    __shared__ signed char rot[SBP_MAXQ];         // rotation bin of an accepted match, -1 otherwise
    __shared__ int hist[30], keep[3];
    const int lane = threadIdx.x;
    for (int w = lane; w < 2048; w += 64) occ[w] = 0;
    if (lane < 30) hist[lane] = 0;
    __syncthreads();
    if (a.t_occ) for (int j = lane; j < a.nt; j += 64) if (a.t_occ[j]) atomicOr(&occ[j >> 5], 1u << (j & 31));
    __syncthreads();
    const float factor = 1.0f / 30;
    int nm = 0;
    for (int i = 0; i < a.nq; i++) {
        const int total = a.cnt[i], navail = total < SBP_K ? total : SBP_K;
        unsigned long long key = ~0ull; bool fr = false;
        if (lane < navail) { key = a.keys[(size_t)i * SBP_K + lane]; const int j = (int)(key & 0xFFFF); fr = !((occ[j >> 5] >> (j & 31)) & 1u); }
        const unsigned long long fm = __ballot(fr);
        unsigned long long k1 = ~0ull, k2 = ~0ull;
        const bool rescan = total > SBP_K && (a.map_mode ? __popcll(fm) < 2 : fm == 0);
        if (rescan) sbp_rescan(a, i, occ, lane, k1, k2);
        else if (fm) {
            k1 = __shfl(key, __ffsll((long long)fm) - 1);
            const unsigned long long f2 = fm & (fm - 1);
            if (f2) k2 = __shfl(key, __ffsll((long long)f2) - 1);
        }
        int mi = -1, md = 256, bin = -1;
        if (k1 != ~0ull) {
            const int j = (int)(k1 & 0xFFFF), d = (int)(k1 >> 32);
            bool acc = d < 256 && d <= a.th_high;          // bestDist starts at 256 and only strictly smaller distances enter
            if (acc && a.map_mode) {                       // ORBmatcher.cc:117-124: same octave && best > ratio * second -> no match
                int d2 = 256, lvl2 = -1;
                if (k2 != ~0ull && (int)(k2 >> 32) < 256) { d2 = (int)(k2 >> 32); lvl2 = a.t_kp[(int)(k2 & 0xFFFF)].octave; }
                if (a.t_kp[j].octave == lvl2 && (float)d > __fmul_rn(a.nn_ratio, (float)d2)) acc = false;
            }
            if (acc) {
                mi = j; md = d; nm++;
                if (a.q_blocks[i] && lane == 0) occ[j >> 5] |= 1u << (j & 31);
                if (a.check_orientation && !a.map_mode) {
                    float r = __fsub_rn(a.q_angle[i], a.t_kp[j].angle);
                    if (r < 0.0f) r = __fadd_rn(r, 360.0f);
                    bin = (int)roundf(__fmul_rn(r, factor));
                    if (bin == 30) bin = 0;
                }
            }
        }
        if (lane == 0) { a.match_idx[i] = mi; a.match_dist[i] = md; rot[i] = (signed char)bin; }
        __syncthreads();                                   // the occupancy bit before the next query reads it
    }
    if (a.check_orientation && !a.map_mode) {              // ComputeThreeMaxima (ORBmatcher.cc:1630-1673) + cull (1473-1487)
        for (int i = lane; i < a.nq; i += 64) if (rot[i] >= 0) atomicAdd(&hist[rot[i]], 1);
        __syncthreads();
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int b = 0; b < 30; b++) {
                const int s = hist[b];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
                else if (s > max3) { max3 = s; ind3 = b; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) ind3 = -1;
            keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
        }
        __syncthreads();
        int gone = 0;
        for (int i = lane; i < a.nq; i += 64) {
            const int b = rot[i];
            if (b >= 0 && b != keep[0] && b != keep[1] && b != keep[2]) { a.match_idx[i] = -1; gone++; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gone += __shfl_xor(gone, o);
        nm -= gone;
    }
    if (lane == 0) *a.n_matches = nm;
}

size_t match_sbp_scratch_bytes(int nq) { return AL((size_t)nq * SBP_K, unsigned long long) + AL(nq, int) + 64; }

// device-resident guided search.  a.keys / a.cnt are carved from `scratch` (match_sbp_scratch_bytes(nq) bytes).
int match_sbp_enqueue(hipStream_t st, SbpDev a, void *scratch)
{
    if (a.nq < 1) return HVO_OK;
    if (a.nq > SBP_MAXQ || a.nt > 65535) return HVO_ERR_UNSUPPORTED;
    char *s = (char *)scratch;
    a.keys = (unsigned long long *)s; s += AL((size_t)a.nq * SBP_K, unsigned long long);
    a.cnt = (int *)s;
    hipLaunchKernelGGL(k_search_by_projection, dim3(a.nq), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_sbp_epilogue, dim3(1), dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

// host-array form (hvo_search_by_projection / hvo_search_by_projection_map): stage, run, fetch
// ------------------------------------------------------------------------------------------------
// The projection prologues of the two guided searches, on the device: what used to cross PCIe per frame as six query arrays is
// computed where the search reads it.
//   k_project_last   ORBmatcher::SearchByProjection(Cur, Last): x3Dc = Rcw x3Dw + tcw, u, v, the bounds tests, radius, octave band, ur
//                    (src/ORBmatcher.cc:1381-1405; cv::Mat arithmetic as oracle/match.c orc_project_last states it)
//   k_track_windows  SearchByProjection(F, vpMapPoints, th): RadiusByViewingCos, th, scale[level], band [level - 1, level] (55-70, 134-140)
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float gemm3_row(const float *a, float b0, float b1, float b2, float c)
{
    float t = __fmul_rn(a[0], b0); t = __fadd_rn(t, __fmul_rn(a[1], b1)); t = __fadd_rn(t, __fmul_rn(a[2], b2));
    return (float)((double)t * 1.0 + (double)c * 1.0);
}
__global__ __launch_bounds__(256) void k_project_last(ProjDev P, int n, const float *__restrict__ x3Dw, const int *__restrict__ q_index,
                                                      const hvo_keypoint *__restrict__ last_kp, float *__restrict__ q_u, float *__restrict__ q_v,
                                                      float *__restrict__ q_radius, int *__restrict__ q_min, int *__restrict__ q_max, float *__restrict__ q_ur)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u = 1e30f, v = 1e30f, radius = 0.f, ur = 0.f; int lo = 0, hi = -1;       // a point that fails a test: no grid cell, never searched
    const float X = x3Dw[3 * i], Y = x3Dw[3 * i + 1], Z = x3Dw[3 * i + 2];
    const float xc = gemm3_row(P.Rcw, X, Y, Z, P.tcw[0]), yc = gemm3_row(P.Rcw + 3, X, Y, Z, P.tcw[1]), zc = gemm3_row(P.Rcw + 6, X, Y, Z, P.tcw[2]);
    const float invzc = (float)(1.0 / (double)zc);
    if (!(invzc < 0)) {
        const float uu = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, xc), invzc), P.cx), vv = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, yc), invzc), P.cy);
        if (!(uu < P.mnMinX || uu > P.mnMaxX) && !(vv < P.mnMinY || vv > P.mnMaxY)) {
            const int oct = last_kp[q_index[i]].octave;
            u = uu; v = vv; radius = __fmul_rn(P.th, P.sf[oct]);
            if (P.fwd) { lo = oct; hi = -1; } else if (P.bwd) { lo = 0; hi = oct; } else { lo = oct - 1; hi = oct + 1; }
            ur = __fsub_rn(uu, __fmul_rn(P.mbf, invzc));
        }
    }
    q_u[i] = u; q_v[i] = v; q_radius[i] = radius; q_min[i] = lo; q_max[i] = hi; q_ur[i] = ur;
}
void match_project_setup(ProjDev &P, const float *Tcw, const float *Tlw, float mb, int mono)
{
    // twc = -Rcw.t() * tcw (double sums: the transposed product takes the generic gemm), tlc = Rlw * twc + tlw (the small-matrix path)
    const float Rcw[9] = { Tcw[0], Tcw[1], Tcw[2], Tcw[4], Tcw[5], Tcw[6], Tcw[8], Tcw[9], Tcw[10] }, tcw[3] = { Tcw[3], Tcw[7], Tcw[11] };
    const float Rlw[9] = { Tlw[0], Tlw[1], Tlw[2], Tlw[4], Tlw[5], Tlw[6], Tlw[8], Tlw[9], Tlw[10] }, tlw[3] = { Tlw[3], Tlw[7], Tlw[11] };
    float twc[3], tlc[3];
    for (int r = 0; r < 3; r++) { double s0 = 0; for (int k = 0; k < 3; k++) s0 += (double)Rcw[3 * k + r] * (double)tcw[k]; twc[r] = (float)(s0 * -1.0); }
    for (int r = 0; r < 3; r++) { float t = Rlw[3 * r] * twc[0]; t += Rlw[3 * r + 1] * twc[1]; t += Rlw[3 * r + 2] * twc[2]; tlc[r] = (float)((double)t * 1.0 + (double)tlw[r] * 1.0); }
    for (int q = 0; q < 9; q++) P.Rcw[q] = Rcw[q];
    for (int q = 0; q < 3; q++) P.tcw[q] = tcw[q];
    P.fwd = (tlc[2] > mb && !mono) ? 1 : 0; P.bwd = (-tlc[2] > mb && !mono) ? 1 : 0;
}
int match_project_last_enqueue(hipStream_t st, const ProjDev &P, int n, const float *d_x3Dw, const int *d_qidx, const hvo_keypoint *d_last_kp,
                               float *q_u, float *q_v, float *q_radius, int *q_min, int *q_max, float *q_ur)
{
    if (n < 1) return HVO_OK;
    hipLaunchKernelGGL(k_project_last, dim3((n + 255) / 256), dim3(256), 0, st, P, n, d_x3Dw, d_qidx, d_last_kp, q_u, q_v, q_radius, q_min, q_max, q_ur);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}
__global__ __launch_bounds__(256) void k_track_windows(int n, const int *__restrict__ level, const float *__restrict__ view_cos, float th, int bfactor, ProjDev P,
                                                       float *__restrict__ q_radius, int *__restrict__ q_min, int *__restrict__ q_max)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r = (double)view_cos[i] > 0.998 ? 2.5f : 4.0f;
    if (bfactor) r = __fmul_rn(r, th);
    const int l = level[i];
    q_radius[i] = __fmul_rn(r, P.sf[l]); q_min[i] = l - 1; q_max[i] = l;
}

// SearchByProjection(F, vpMapPoints, th) from the tracker's own per-point fields (mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos)
int match_search_by_projection_tracked(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *proj_x, const float *proj_y, const float *proj_xr,
                                       const int32_t *level, const float *view_cos, const uint8_t *q_blocks, float th,
                                       const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                                       float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, float nn_ratio,
                                       int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (nq > SBP_MAXQ || nt > 65535) return HVO_ERR_UNSUPPORTED;
    const size_t in_bytes = AL(nq * 32, char) + 9 * AL(nq, float) + AL(nq, char) + AL(nt, hvo_keypoint) + AL(nt, float) + AL(nt, char) + AL(nt * 32, char);
    const size_t out_bytes = AL(2 * (size_t)nq + 1, int32_t);
    int rc = arena_begin(ctx, in_bytes + out_bytes + match_sbp_scratch_bytes(nq) + 1024, in_bytes + out_bytes + 1024);
    if (rc) return rc;
    SbpDev a; memset(&a, 0, sizeof(a));
    a.q_desc = arena_up(ctx, q_desc, (size_t)nq * 32); a.q_desc_index = nullptr;
    a.q_u = arena_up(ctx, proj_x, (size_t)nq); a.q_v = arena_up(ctx, proj_y, (size_t)nq); a.q_ur = arena_up(ctx, proj_xr, (size_t)nq);
    const int *d_level = arena_up(ctx, level, (size_t)nq); const float *d_vc = arena_up(ctx, view_cos, (size_t)nq);
    float *d_radius = arena_dev<float>(ctx, nq); int *d_min = arena_dev<int>(ctx, nq), *d_max = arena_dev<int>(ctx, nq);
    ProjDev P; memset(&P, 0, sizeof(P));
    for (int l = 0; l < HVO_MAX_LEVELS; l++) P.sf[l] = ctx->scale[l];
    hipLaunchKernelGGL(k_track_windows, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, nq, d_level, d_vc, th, th != 1.0f ? 1 : 0, P, d_radius, d_min, d_max);
    a.q_radius = d_radius; a.q_min_level = d_min; a.q_max_level = d_max; a.q_angle = nullptr; a.q_blocks = arena_up(ctx, q_blocks, (size_t)nq);
    a.t_kp = arena_up(ctx, t_kp, (size_t)nt); a.t_uright = arena_up(ctx, t_uright, (size_t)nt); a.t_occ = arena_up(ctx, t_occupied, (size_t)nt);
    a.t_desc = arena_up(ctx, t_desc, (size_t)nt * 32);
    a.nq = nq; a.nt = nt; a.mnMinX = mnMinX; a.mnMinY = mnMinY; a.mnMaxX = mnMaxX; a.mnMaxY = mnMaxY;
    a.th_high = th_high; a.check_orientation = 0; a.map_mode = 1; a.nn_ratio = nn_ratio;
    int32_t *dout = arena_dev<int32_t>(ctx, 2 * (size_t)nq + 1), *hout = arena_host<int32_t>(ctx, 2 * (size_t)nq + 1);
    a.match_idx = dout; a.match_dist = dout + nq; a.n_matches = dout + 2 * nq;
    void *scratch = arena_dev<char>(ctx, match_sbp_scratch_bytes(nq));
    if ((rc = match_sbp_enqueue(ctx->stream, a, scratch))) { ctx->last_error = "guided search launch"; return rc; }
    HVO_HIP(hipMemcpyAsync(hout, dout, (2 * (size_t)nq + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(match_idx, hout, (size_t)nq * sizeof(int32_t)); memcpy(match_dist, hout + nq, (size_t)nq * sizeof(int32_t));
    *n_matches = hout[2 * nq];
    return HVO_OK;
}

int match_search_by_projection(hvo_ctx *ctx, const uint8_t *q_desc, int nq, const float *q_u, const float *q_v, const float *q_radius,
                               const int32_t *q_min_level, const int32_t *q_max_level, const float *q_ur, const float *q_angle, const uint8_t *q_blocks,
                               const hvo_keypoint *t_kp, const float *t_uright, const uint8_t *t_occupied, const uint8_t *t_desc, int nt,
                               float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, int th_high, int check_orientation, int map_mode, float nn_ratio,
                               int32_t *match_idx, int32_t *match_dist, int *n_matches)
{
    if (nq > SBP_MAXQ || nt > 65535) return HVO_ERR_UNSUPPORTED;
    const size_t in_bytes = AL(nq * 32, char) + 7 * AL(nq, float) + AL(nq, char) + AL(nt, hvo_keypoint) + AL(nt, float) + AL(nt, char) + AL(nt * 32, char);
    const size_t out_bytes = AL(2 * (size_t)nq + 1, int32_t);
    int rc = arena_begin(ctx, in_bytes + out_bytes + match_sbp_scratch_bytes(nq) + 1024, in_bytes + out_bytes + 1024);
    if (rc) return rc;
    SbpDev a; memset(&a, 0, sizeof(a));
    a.q_desc = arena_up(ctx, q_desc, (size_t)nq * 32); a.q_desc_index = nullptr;
    a.q_u = arena_up(ctx, q_u, (size_t)nq); a.q_v = arena_up(ctx, q_v, (size_t)nq); a.q_radius = arena_up(ctx, q_radius, (size_t)nq);
    a.q_min_level = arena_up(ctx, q_min_level, (size_t)nq); a.q_max_level = arena_up(ctx, q_max_level, (size_t)nq);
    a.q_ur = arena_up(ctx, q_ur, (size_t)nq); a.q_angle = arena_up(ctx, q_angle, (size_t)nq); a.q_blocks = arena_up(ctx, q_blocks, (size_t)nq);
    a.t_kp = arena_up(ctx, t_kp, (size_t)nt); a.t_uright = arena_up(ctx, t_uright, (size_t)nt); a.t_occ = arena_up(ctx, t_occupied, (size_t)nt);
    a.t_desc = arena_up(ctx, t_desc, (size_t)nt * 32);
    a.nq = nq; a.nt = nt; a.mnMinX = mnMinX; a.mnMinY = mnMinY; a.mnMaxX = mnMaxX; a.mnMaxY = mnMaxY;
    a.th_high = th_high; a.check_orientation = check_orientation && q_angle; a.map_mode = map_mode; a.nn_ratio = nn_ratio;
    int32_t *dout = arena_dev<int32_t>(ctx, 2 * (size_t)nq + 1), *hout = arena_host<int32_t>(ctx, 2 * (size_t)nq + 1);
    a.match_idx = dout; a.match_dist = dout + nq; a.n_matches = dout + 2 * nq;
    void *scratch = arena_dev<char>(ctx, match_sbp_scratch_bytes(nq));
    if ((rc = match_sbp_enqueue(ctx->stream, a, scratch))) { ctx->last_error = "guided search launch"; return rc; }
    HVO_HIP(hipMemcpyAsync(hout, dout, (2 * (size_t)nq + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(match_idx, hout, (size_t)nq * sizeof(int32_t)); memcpy(match_dist, hout + nq, (size_t)nq * sizeof(int32_t));
    *n_matches = hout[2 * nq];
    return HVO_OK;
}

__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const hvo_keypoint *__restrict__ kp, const hvo_keypoint *__restrict__ kpun, const int *__restrict__ n_ptr, int n_fixed,
                                                          const uint16_t *__restrict__ depth, int pitch, int w, int h, float dfac, float bf,
                                                          float *__restrict__ uright, float *__restrict__ zdepth)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_ptr ? *n_ptr : n_fixed;
    if (i >= n) return;
    float ur = -1.f, z = -1.f;
    const int v = (int)kp[i].y, u = (int)kp[i].x;            // imDepth.at<float>(v, u) with float v,u (Frame.cc:1950-1953)
    if (u >= 0 && v >= 0 && u < w && v < h) {
        const float d = __fmul_rn((float)depth[(size_t)v * pitch + u], dfac);
        if (d > 0 && (double)d < 7.0) { z = d; ur = __fsub_rn(kpun[i].x, __fdiv_rn(bf, d)); }
    }
    uright[i] = ur; zdepth[i] = z;
}

// device-resident Frame::ComputeStereoFromRGBD: n is read from *d_n when d_n is not null (the key-point count of the frame
// is only known on the device), else n_max is the count; the grid covers n_max
int match_stereo_enqueue(hipStream_t st, const hvo_keypoint *d_kp, const hvo_keypoint *d_kpun, const int *d_n, int n_max, const uint16_t *d_depth, int pitch,
                         int w, int h, float dfac, float bf, float *d_uright, float *d_zdepth)
{
    if (n_max < 1) return HVO_OK;
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((n_max + 255) / 256), dim3(256), 0, st, d_kp, d_kpun, d_n, n_max, d_depth, pitch, w, h, dfac, bf, d_uright, d_zdepth);
    return hipGetLastError() == hipSuccess ? HVO_OK : HVO_ERR_HIP;
}

int match_stereo_from_rgbd(hvo_ctx *ctx, const hvo_keypoint *kp, const hvo_keypoint *kpun, int n, const uint16_t *depth, int w, int h, int stride,
                           float bf, float *uright, float *zdepth)
{
    const size_t db = (size_t)w * h * 2;
    int rc = arena_begin(ctx, 2 * AL(n, hvo_keypoint) + AL(db, char) + 2 * AL(n, float), 2 * AL(n, hvo_keypoint) + AL(db, char) + 2 * AL(n, float));
    if (rc) return rc;
    const hvo_keypoint *dk = arena_up(ctx, kp, (size_t)n), *dku = arena_up(ctx, kpun, (size_t)n);
    uint16_t *hd = arena_host<uint16_t>(ctx, (size_t)w * h), *dd = arena_dev<uint16_t>(ctx, (size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(hd + (size_t)y * w, (const char *)depth + (size_t)y * stride, (size_t)w * 2);
    HVO_HIP(hipMemcpyAsync(dd, hd, db, hipMemcpyHostToDevice, ctx->stream));
    float *dur = arena_dev<float>(ctx, (size_t)n), *dz = arena_dev<float>(ctx, (size_t)n);
    float *hur = arena_host<float>(ctx, (size_t)n), *hz = arena_host<float>(ctx, (size_t)n);
    if ((rc = match_stereo_enqueue(ctx->stream, dk, dku, nullptr, n, dd, w, w, h, ctx->p.depth_map_factor, bf, dur, dz))) return rc;
    HVO_HIP(hipMemcpyAsync(hur, dur, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipMemcpyAsync(hz, dz, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(uright, hur, (size_t)n * 4); memcpy(zdepth, hz, (size_t)n * 4);
    return HVO_OK;
}

// diagnostics (tools/match_rate.py; not part of include/hvo.h): device time of `iters` back-to-back launches of the knn-2 (kind 0) or
// distance-matrix (kind 1) kernel on descriptors resident in HBM -> ms per launch
extern "C" int hvo_debug_match_rate(hvo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int kind, int iters, float *ms)
{
    if (!ctx || !q || !t || !ms || nq < 1 || nt < 1 || iters < 1 || nt > 65535) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    const size_t ob = kind == 0 ? (size_t)nq * 4 * sizeof(int32_t) : (size_t)nq * nt * sizeof(uint16_t);
    int rc = arena_begin(ctx, AL(nq * 32, char) + AL(nt * 32, char) + AL(ob, char), AL(nq * 32, char) + AL(nt * 32, char));
    if (rc) return rc;
    const uint8_t *dq = arena_up(ctx, q, (size_t)nq * 32), *dt = arena_up(ctx, t, (size_t)nt * 32);
    char *dout = arena_dev<char>(ctx, ob);
    hipEvent_t e0, e1;
    HVO_HIP(hipEventCreate(&e0)); HVO_HIP(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; pass++) {                          // pass 0 warms up
        HVO_HIP(hipEventRecord(e0, ctx->stream));
        for (int i = 0; i < iters; i++) {
            if (kind == 0) hipLaunchKernelGGL(k_hamming_knn2, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, (const ulonglong4 *)dq, nq, (const ulonglong4 *)dt, nt,
                                              (int32_t *)dout, (int32_t *)dout + (size_t)nq * 2);
            else hipLaunchKernelGGL(k_hamming_matrix, dim3(std::min((nt + 63) / 64, 64), (nq + 3) / 4), dim3(256), 0, ctx->stream, (const ulonglong4 *)dq, nq,
                                    (const ulonglong4 *)dt, nt, (uint16_t *)dout);
        }
        HVO_HIP(hipEventRecord(e1, ctx->stream));
        HVO_HIP(hipEventSynchronize(e1));
    }
    float tot = 0;
    HVO_HIP(hipEventElapsedTime(&tot, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms = tot / iters;
    return HVO_OK;
}

#include "line_track.inc"

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

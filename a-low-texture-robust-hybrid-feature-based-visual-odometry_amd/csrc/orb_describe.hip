// orb_describe.hip -- orientation and descriptor of the selected key points (the tail of the "pyramid+BRIEF pass").
//
//   k_moments    IC_Angle's patch moments m01, m10          ORBextractor.cc:75-102     two key points per wave, a lane = one patch row
//   k_kpfinish   fastAtan2, cos / sin of the angle, record  ORBextractor.cc:470-477, 112-114, 1093-1099   a lane = one key point
//   k_brief      steered BRIEF-256                          ORBextractor.cc:106-145    a wave = one key point, a lane = 4 tests
//
// Why three kernels: an instruction costs a wave the same issue slots whether one lane or 64 execute it, so what is per KEY POINT and
// expensive (the float polynomial of fastAtan2, cos / sin in double -- the oracle rounds the double result to float) runs with one key
// point per LANE (k_kpfinish: ~3 instructions per key point instead of ~180 per wave), and only what is per PIXEL keeps a wave per
// key point.  The moments read a patch row with two 16-byte loads and reduce it with masked v_dot4_u32_u8 (weights u + 16) instead
// of a 16-step byte loop; rows are summed by DPP.  Work is indexed by (level, slot) from a static chunk table -- a workgroup's level is
// uniform -- and the compacted key-point index is the running sum of the levels' counts (level-major order of ORBextractor.cc:1041-1103).
// All workgroups of a frame are dealt to one XCD (blocks b, b + 8, ...) so the frame's level images are fetched into one L2.
// Between the kernels the moments, then (cos, sin), are parked in the key point's own 32-byte descriptor slot.
#include "hvo_internal.hpp"
#include <math.h>
#include <vector>

static __device__ __forceinline__ int od_x(uint32_t c) { return c & 0xFFF; }
static __device__ __forceinline__ int od_y(uint32_t c) { return (c >> 12) & 0xFFF; }
static __device__ __forceinline__ int od_s(uint32_t c) { return c >> 24; }

static __device__ __forceinline__ float od_fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2 (OpenCV 3.2) -- float polynomial, evaluated without contraction
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y), r, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        r = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        r = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) r = __fsub_rn(180.f, r);
    if (y < 0) r = __fsub_rn(360.f, r);
    return r;
}

#define BR_R 18                      // reach of the rotated pattern
#define BR_ROWS (2 * BR_R + 1)       // 37 window rows
#define BR_PITCH 48                  // three 16-byte chunks: columns cx - 19 .. cx + 28

struct DescArgs {
    const uint8_t *pyr0; size_t stride0;               // level 0 of the chunk's first frame (the per-frame input slab)
    const uint8_t *lvl; size_t lvl_stride;             // levels >= 1 (scratch, chunk-local frames)
    const uint8_t *blur; size_t blur_stride;           // all levels blurred (scratch)
    const LevelGeom *lev; int nlevels;
    const int2 *chunks; int nchunks, nframes;
    const uint32_t *lvl_kp; const int *lvl_cnt; int kp_total;
    const int *umax; const int8_t *pattern;
    hvo_keypoint *kp; int *nkp; int cap; uint8_t *desc;
};

// workgroup -> (frame, chunk): blocks b and b + 8 share an XCD
static __device__ __forceinline__ bool od_locate(const DescArgs &A, int &frame, int &level, int &k0, int &cnt, int &base)
{
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int fr8 = q / A.nchunks, ch = q - fr8 * A.nchunks;
    frame = fr8 * 8 + xcd;
    if (frame >= A.nframes) return false;
    const int2 c = A.chunks[ch];
    level = c.x; k0 = c.y;
    const int *lc = A.lvl_cnt + (size_t)frame * A.nlevels;
    cnt = lc[level];
    base = 0;
    for (int l = 0; l < level; l++) base += lc[l];
    return k0 < cnt;
}

template <int CTRL, int ROWMASK>
static __device__ __forceinline__ int od_dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, true);
}

__global__ __launch_bounds__(256) void k_moments(const DescArgs A)
{
    int frame, level, k0, cnt, base;
    if (!od_locate(A, frame, level, k0, cnt, base)) return;
    const LevelGeom L = A.lev[level];
    const uint8_t *img = level == 0 ? A.pyr0 + (size_t)frame * A.stride0 : A.lvl + (size_t)frame * A.lvl_stride + L.lvl_off;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, row = lane & 31;
    const int v = row - 15;
    const bool rowok = row < 31;
    // per-lane constants: the row's bytes x-16 .. x+15 are u = -16 .. 15; |u| <= umax[|v|] stays, weight u + 16
    const int d = A.umax[rowok ? (v < 0 ? -v : v) : 0];
    unsigned msk[8], wgt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        unsigned m = 0, w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int u = 4 * k + b - 16;
            if (rowok && u >= -d && u <= d) m |= 0xFFu << (8 * b);
            w |= (unsigned)(u + 16) << (8 * b);
        }
        msk[k] = m; wgt[k] = w;
    }
    const uint32_t *lk = A.lvl_kp + (size_t)frame * A.kp_total + L.kp_off;
    typedef uint32_t od_u4 __attribute__((ext_vector_type(4), aligned(1)));
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
        const int kk = k0 + rep * 8 + wv * 2 + half;
        const bool valid = kk < cnt;
        const uint32_t c = lk[valid ? kk : cnt - 1];
        const int x = od_x(c) + L.minBX, y = od_y(c) + L.minBY;
        // FAST's region keeps the patch inside the image: 16 <= x < w - 16, rows y - 15 .. y + 15
        const uint8_t *p = img + (size_t)(y + (rowok ? v : 0)) * L.pitch + x - 16;
        const od_u4 q0 = *reinterpret_cast<const od_u4 *>(p), q1 = *reinterpret_cast<const od_u4 *>(p + 16);
        const unsigned px[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
        unsigned s = 0, wd = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned t = px[k] & msk[k];
            s = __builtin_amdgcn_udot4(t, 0x01010101u, s, false);
            wd = __builtin_amdgcn_udot4(t, wgt[k], wd, false);
        }
        int m10 = (int)wd - 16 * (int)s, m01 = v * (int)s;
        // sum over the 32 lanes of the key point: quads, half rows, rows (DPP), then row 0 -> row 1 / row 2 -> row 3
        m10 = od_dpp_add<0xB1, 0xF>(m10); m01 = od_dpp_add<0xB1, 0xF>(m01);
        m10 = od_dpp_add<0x4E, 0xF>(m10); m01 = od_dpp_add<0x4E, 0xF>(m01);
        m10 = od_dpp_add<0x141, 0xF>(m10); m01 = od_dpp_add<0x141, 0xF>(m01);
        m10 = od_dpp_add<0x140, 0xF>(m10); m01 = od_dpp_add<0x140, 0xF>(m01);
        m10 = od_dpp_add<0x142, 0xA>(m10); m01 = od_dpp_add<0x142, 0xA>(m01);     // rows 1 and 3 += lane 15 of the row before
        if (valid && row == 31) *reinterpret_cast<int2 *>(A.desc + ((size_t)frame * A.cap + base + kk) * 32) = make_int2(m01, m10);
    }
}

// one lane per (level, slot): angle, (cos, sin), the key-point record
__global__ __launch_bounds__(256) void k_kpfinish(const DescArgs A)
{
    const int frame = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int *lc = A.lvl_cnt + (size_t)frame * A.nlevels;
    int level = -1, base = 0, acc = 0;
    for (int l = 0; l < A.nlevels; l++) {
        const LevelGeom &G = A.lev[l];
        if (s >= G.kp_off && s < G.kp_off + G.kp_cap) { level = l; base = acc; }
        acc += lc[l];
    }
    if (s == 0) A.nkp[frame] = min(acc, A.cap);
    if (level < 0) return;
    const LevelGeom L = A.lev[level];
    const int k = s - L.kp_off;
    if (k >= lc[level]) return;
    const int idx = base + k;
    const uint32_t c = A.lvl_kp[(size_t)frame * A.kp_total + s];
    uint8_t *slot = A.desc + ((size_t)frame * A.cap + idx) * 32;
    const int2 m = *reinterpret_cast<const int2 *>(slot);
    const int x = od_x(c) + L.minBX, y = od_y(c) + L.minBY;
    hvo_keypoint kp;
    kp.angle = od_fast_atan2_deg((float)m.x, (float)m.y);
    // float angle = kpt.angle * factorPI; a = (float)cos(angle), b = (float)sin(angle)   (ORBextractor.cc:112-114)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float angle = __fmul_rn(kp.angle, factorPI);
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    *reinterpret_cast<float2 *>(slot) = make_float2(a, b);
    const float kx = (float)x, ky = (float)y;
    kp.x = level != 0 ? __fmul_rn(kx, L.scale) : kx;              // keypoint->pt *= scale (ORBextractor.cc:1093-1099)
    kp.y = level != 0 ? __fmul_rn(ky, L.scale) : ky;
    kp.size = (float)L.scaled_patch;
    kp.response = (float)od_s(c);
    kp.octave = level; kp.class_id = -1;
    A.kp[(size_t)frame * A.cap + idx] = kp;
}

__global__ __launch_bounds__(256) void k_brief(const DescArgs A)
{
    int frame, level, k0, cnt, base;
    if (!od_locate(A, frame, level, k0, cnt, base)) return;
    const LevelGeom L = A.lev[level];
    const uint8_t *img = A.blur + (size_t)frame * A.blur_stride + L.img_off;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int4 praw = *reinterpret_cast<const int4 *>(A.pattern + 16 * lane);
    const int8_t *pp = reinterpret_cast<const int8_t *>(&praw);
    float fx[8], fy[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { fx[e] = (float)pp[2 * e]; fy[e] = (float)pp[2 * e + 1]; }
    const uint32_t *__restrict__ lk = A.lvl_kp + (size_t)frame * A.kp_total + L.kp_off;
    uint8_t *slots = A.desc + ((size_t)frame * A.cap + base) * 32;
    // the wave's eight key points: packed position and (cos, sin) of all of them are requested before the first is used
    uint32_t cs[8]; float2 abv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int kk = min(k0 + wv * 8 + i, cnt - 1);
        cs[i] = lk[kk];
        abv[i] = *reinterpret_cast<const float2 *>(slots + (size_t)kk * 32);
    }
    // The 512 samples of a key point lie in a 37 x 37 window of the blurred level.  Gathered straight from global memory they cost the
    // texture addresser a pass per distinct line of every load instruction (8 instructions x ~37 lines); instead the window's rows are
    // fetched with two 16-byte loads per lane (39 rows x 3 chunks), parked in LDS, and the gather runs there.  Double buffered: the
    // next key point's rows are in flight while this one's tests run.  Key points within 19 pixels of the border (reflection) gather from
    // global memory as before.
    __shared__ __attribute__((aligned(16))) uint8_t patch_[4][2][BR_ROWS * BR_PITCH];
    typedef uint32_t od_u4 __attribute__((ext_vector_type(4), aligned(1)));
    auto fetch = [&](int i, uint4 &r0, uint4 &r1) {
        const uint32_t c = cs[i];
        const int cx = od_x(c) + L.minBX, cy = od_y(c) + L.minBY;
        const bool inner = cx >= 19 && cy >= 19 && cx < L.w - 19 && cy < L.h - 19;
        r0 = r1 = make_uint4(0, 0, 0, 0);
        if (inner) {
            const int i0 = lane, i1 = lane + 64;                      // item = row * 3 + chunk
            const int y0 = i0 / 3, c0 = i0 - 3 * y0, y1 = i1 / 3, c1 = i1 - 3 * y1;
            const uint8_t *org = img + (cy - BR_R) * L.pitch + cx - BR_R - 1;
            const od_u4 q0 = *reinterpret_cast<const od_u4 *>(org + y0 * L.pitch + 16 * c0);
            r0 = make_uint4(q0.x, q0.y, q0.z, q0.w);
            if (y1 < BR_ROWS) { const od_u4 q1 = *reinterpret_cast<const od_u4 *>(org + y1 * L.pitch + 16 * c1); r1 = make_uint4(q1.x, q1.y, q1.z, q1.w); }
        }
    };
    uint4 r0, r1;
    if (k0 + wv * 8 < cnt) fetch(0, r0, r1);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int kk = k0 + wv * 8 + i;
        if (kk >= cnt) break;                                      // wave-uniform
        const uint32_t c = cs[i];
        const int cx = od_x(c) + L.minBX, cy = od_y(c) + L.minBY;
        const float a = abv[i].x, b = abv[i].y;
        // the rotated pattern reaches at most 18 pixels (|p| <= 13 per axis: 13 * sqrt 2 = 18.4): away from the border no reflection
        const bool inner = cx >= 19 && cy >= 19 && cx < L.w - 19 && cy < L.h - 19;     // wave-uniform
        uint8_t *pb = patch_[wv][i & 1];
        if (inner) {
            reinterpret_cast<uint4 *>(pb)[lane] = r0;
            if (lane + 64 < BR_ROWS * 3) reinterpret_cast<uint4 *>(pb)[lane + 64] = r1;
        }
        if (i + 1 < 8 && kk + 1 < cnt) fetch(i + 1, r0, r1);
        int tv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int ry = __float2int_rn(__fadd_rn(__fmul_rn(fx[e], b), __fmul_rn(fy[e], a)));
            const int rx = __float2int_rn(__fsub_rn(__fmul_rn(fx[e], a), __fmul_rn(fy[e], b)));
            if (inner) tv[e] = pb[(ry + BR_R) * BR_PITCH + rx + BR_R + 1];
            else {
                int yy = cy + ry, xx = cx + rx;
                yy = yy < 0 ? -yy : (yy >= L.h ? 2 * (L.h - 1) - yy : yy);
                xx = xx < 0 ? -xx : (xx >= L.w ? 2 * (L.w - 1) - xx : xx);
                tv[e] = img[yy * L.pitch + xx];
            }
        }
        unsigned nib = 0;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) nib |= (unsigned)(tv[2 * qq] < tv[2 * qq + 1]) << qq;
        // bytes on even lanes, dwords on lanes 8w: row_shl reads lane + n of the 16-lane row
        const unsigned byte = nib | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x101, 0xF, 0xF, true) << 4);
        const unsigned w = byte | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)byte, 0x102, 0xF, 0xF, true) << 8) |
                           ((unsigned)__builtin_amdgcn_update_dpp(0, (int)byte, 0x104, 0xF, 0xF, true) << 16) |
                           ((unsigned)__builtin_amdgcn_update_dpp(0, (int)byte, 0x106, 0xF, 0xF, true) << 24);
        if ((lane & 7) == 0) reinterpret_cast<uint32_t *>(slots + (size_t)kk * 32)[lane >> 3] = w;
    }
}

// host ------------------------------------------------------------------------------------------------------------------
int orb_describe_build(hvo_ctx *ctx)
{
    OrbPlan &P = ctx->orb;
    std::vector<int2> ch;
    for (int l = 0; l < P.nlevels; l++)
        for (int k0 = 0; k0 < P.lev[l].kp_cap; k0 += 32) ch.push_back(make_int2(l, k0));
    P.n_kpchunks = (int)ch.size();
    if (hipMalloc((void **)&P.d_kpchunks, ch.size() * sizeof(int2)) != hipSuccess) { ctx->last_error = "hipMalloc(kpchunks)"; return HVO_ERR_HIP; }
    HVO_HIP(hipMemcpy(P.d_kpchunks, ch.data(), ch.size() * sizeof(int2), hipMemcpyHostToDevice));
    return HVO_OK;
}

int orb_describe_run(hvo_ctx *ctx, int c0, int n, hipStream_t st)
{
    OrbPlan &P = ctx->orb;
    DescArgs A;
    A.pyr0 = P.d_pyr + (size_t)c0 * P.pyr_bytes; A.stride0 = P.pyr_bytes; A.lvl = P.d_lvl; A.lvl_stride = P.lvl_bytes; A.blur = P.d_blur; A.blur_stride = P.blur_bytes;
    A.lev = P.d_lev; A.nlevels = P.nlevels;
    A.chunks = P.d_kpchunks; A.nchunks = P.n_kpchunks; A.nframes = n;
    A.lvl_kp = P.d_lvl_kp; A.lvl_cnt = P.d_lvl_cnt; A.kp_total = P.kp_total; A.umax = ctx->d_umax; A.pattern = ctx->d_pattern;
    A.kp = P.d_kp + (size_t)c0 * P.kp_cap; A.nkp = P.d_nkp + c0; A.cap = P.kp_cap; A.desc = P.d_desc + (size_t)c0 * P.kp_cap * 32;
    const unsigned nb = (unsigned)((n + 7) / 8 * 8 * P.n_kpchunks);
    int id = hvo_prof_begin(ctx, "orb_orient", st);
    hipLaunchKernelGGL(k_moments, dim3(nb), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_kpfinish, dim3((P.kp_total + 255) / 256, n), dim3(256), 0, st, A);
    hvo_prof_end(ctx, id);
    id = hvo_prof_begin(ctx, "orb_brief", st);
    hipLaunchKernelGGL(k_brief, dim3(nb), dim3(256), 0, st, A);
    hvo_prof_end(ctx, id);
    return HVO_OK;
}

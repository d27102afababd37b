// orb.hip -- batched ORB extraction for gfx950 (MI355X).
//
// Replaces ORBextractor::operator() (reference src/ORBextractor.cc:1041-1103) for a batch of
// independent frames.  Kernels (one launch covers every frame of the batch):
//   k_resize_dw     ComputePyramid                    ORBextractor.cc:1105-1130  (cv::resize u8 bilinear; k_resize = byte-load
//                   formulation for scale factors > 4/3 or HVO_RESIZE_BYTES=1)
//   k_fast_cells    per-cell cv::FAST + fallback      ORBextractor.cc:787-827
//   k_octree        DistributeOctTree / DivideNode    ORBextractor.cc:537-761, 479-535
//   (orientation and descriptors: orb_describe.hip; the fused per-level pass that replaces k_resize / k_fast_cells / k_blur7
//    where the geometry fits its LDS tile: orb_level.hip)
//   k_blur7         GaussianBlur 7x7 sigma 2          ORBextractor.cc:1083-1084
//
// Data layout in HBM (per frame): one pyramid slab holding the nlevels u8 images back to back
// (row pitch = width rounded up to 64 B, no apron: REFLECT_101 is applied by index reflection
// where the reference reads its 19-px apron), a second slab of the same shape for the blurred
// levels, fixed-capacity candidate slabs per FAST cell, and the output keypoint / descriptor
// slabs (cap x 28 B, cap x 32 B).
//
// Bit-exactness notes: all image arithmetic is integer; float steps (fastAtan2 polynomial,
// steered-BRIEF coordinate rotation) use __f*_rn intrinsics so no FMA contraction can occur.
#include "hvo_internal.hpp"
#include <math.h>
#include <string.h>
#include <algorithm>

// =====================================================================================
// device helpers
// =====================================================================================
static __device__ __forceinline__ int reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p < 0 ? 0 : p;
}

static __device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

// =====================================================================================
// K1: pyramid level l from level l-1 (cv::resize INTER_LINEAR u8, 11-bit fixed point)
// =====================================================================================
#define RESIZE_ROWS 16
#ifdef HVO_WPE_RESIZE
__attribute__((amdgpu_waves_per_eu(HVO_WPE_RESIZE)))
#endif
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src_base, size_t src_stride, uint8_t *__restrict__ dst_base, size_t dst_stride,
                                                LevelGeom S, LevelGeom D,
                                                const int *__restrict__ xofs, const int *__restrict__ xalpha,
                                                const int *__restrict__ yofs, const int *__restrict__ ybeta)
{
    const int dx0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (dx0 >= D.w) return;
    const uint8_t *src = src_base + (size_t)blockIdx.z * src_stride;       // (the bases already point at the level inside frame 0's slab)
    uint8_t *dst = dst_base + (size_t)blockIdx.z * dst_stride;
    // horizontal taps of this thread's 4 pixels are row independent: fetch them once
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = min(dx0 + i, D.w - 1);
        sx[i] = xofs[dx];
        const int xa = xalpha[dx];
        a0[i] = (short)(xa & 0xFFFF); a1[i] = (short)(xa >> 16);
        sx1[i] = min(sx[i] + 1, S.w - 1);                       // a1 == 0 whenever sx+1 is outside
    }
    // RESIZE_ROWS rows per thread: the launch is dispatch bound with small workgroups (~6 ns per workgroup)
#pragma unroll 4
    for (int r = 0; r < RESIZE_ROWS; r++) {
        const int dy = (blockIdx.y * blockDim.y + threadIdx.y) * RESIZE_ROWS + r;
        if (dy >= D.h) break;
        const int yo = yofs[dy];
        const int sy0 = yo & 0xFFFF, sy1 = yo >> 16;
        const int yb = ybeta[dy];
        const int b0 = (short)(yb & 0xFFFF), b1 = (short)(yb >> 16);
        const uint8_t *r0 = src + (size_t)sy0 * S.pitch, *r1 = src + (size_t)sy1 * S.pitch;
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (dx0 + i < D.w) {
                const int t0 = r0[sx[i]] * a0[i] + r0[sx1[i]] * a1[i];
                const int t1 = r1[sx[i]] * a0[i] + r1[sx1[i]] * a1[i];
                const int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
                out |= (uint32_t)(v & 0xFF) << (8 * i);
            }
        }
        // pitch is a multiple of 64 so a full dword store stays inside the row
        *reinterpret_cast<uint32_t *>(dst + (size_t)dy * D.pitch + dx0) = out;
    }
}

// Same arithmetic, fewer and wider loads: the 8 source bytes a thread needs from a row (4 pixels x 2 taps) lie
// within 9 bytes of a 4-byte aligned address when the scale factor is <= 4/3 (checked per level on the host),
// so they are fetched as three dwords and picked with v_perm_b32 (selector fixed per thread); the right taps
// come from the same window shifted by one byte.  The horizontal pass of a source row is kept for the next
// output row, which reuses it whenever its upper source row is this row's lower one (5 rows in 6 at 1.2).
__global__ __launch_bounds__(256) void k_resize_dw(const uint8_t *__restrict__ src_base, size_t src_stride, uint8_t *__restrict__ dst_base, size_t dst_stride,
                                                   LevelGeom S, LevelGeom D,
                                                   const int *__restrict__ xofs, const int *__restrict__ xalpha,
                                                   const int *__restrict__ yofs, const int *__restrict__ ybeta)
{
    const int dx0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (dx0 >= D.w) return;
    const uint8_t *src = src_base + (size_t)blockIdx.z * src_stride;
    uint8_t *dst = dst_base + (size_t)blockIdx.z * dst_stride;
    int a0[4], a1[4];
    unsigned sel = 0; int wbase = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = min(dx0 + i, D.w - 1);
        const int sx = xofs[dx];
        const int xa = xalpha[dx];
        a0[i] = (short)(xa & 0xFFFF); a1[i] = (short)(xa >> 16);      // a1 == 0 whenever sx + 1 is outside the image
        if (i == 0) wbase = sx >> 2;
        sel |= (unsigned)(sx - 4 * wbase) << (8 * i);                 // byte offsets 0..7 in the dword pair (host-checked)
    }
    auto hrow = [&](int sy, int t[4]) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(src + (size_t)sy * S.pitch) + wbase;
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];              // the third dword stays inside the pyramid buffer (a later row or level)
        const uint32_t L = __builtin_amdgcn_perm(w1, w0, sel);
        const uint32_t R = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(w2, w1, 1), __builtin_amdgcn_alignbyte(w1, w0, 1), sel);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = (int)((L >> (8 * i)) & 0xFFu) * a0[i] + (int)((R >> (8 * i)) & 0xFFu) * a1[i];
    };
    int prev_sy = -1, pt[4] = { 0, 0, 0, 0 };
#pragma unroll 4
    for (int r = 0; r < RESIZE_ROWS; r++) {
        const int dy = (blockIdx.y * blockDim.y + threadIdx.y) * RESIZE_ROWS + r;
        if (dy >= D.h) break;
        const int yo = yofs[dy];
        const int sy0 = yo & 0xFFFF, sy1 = yo >> 16;
        const int yb = ybeta[dy];
        const int b0 = (short)(yb & 0xFFFF), b1 = (short)(yb >> 16);
        int t0[4], t1[4];
        if (sy0 == prev_sy) {
#pragma unroll
            for (int i = 0; i < 4; i++) t0[i] = pt[i];
        } else hrow(sy0, t0);
        hrow(sy1, t1);
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = (((b0 * (t0[i] >> 4)) >> 16) + ((b1 * (t1[i] >> 4)) >> 16) + 2) >> 2;
            if (dx0 + i < D.w) out |= (uint32_t)(v & 0xFF) << (8 * i);
            pt[i] = t1[i];
        }
        prev_sy = sy1;
        *reinterpret_cast<uint32_t *>(dst + (size_t)dy * D.pitch + dx0) = out;
    }
}

// =====================================================================================
// K2: FAST-9-16 per cell, strict 3x3 NMS inside the cell view, threshold fallback
// =====================================================================================
// FAST corner score (cornerScore<16> of OpenCV): with d[k] = v - ring[k],
//   S = max over the 16 arcs of 9 contiguous ring pixels of max(min d, -max d) - 1
// A pixel is a corner at threshold t  <=>  S >= t, so one threshold-free score map serves both
// the iniThFAST pass and the minThFAST fallback; NMS survivors at t are the strict 8-neighbour
// local maxima of S with S >= t.
static __device__ __forceinline__ int fast_score(const uint8_t *tile, int tp, int x, int y, int v)
{
    const uint8_t *p = tile + y * tp + x;
    int d[16];
    d[0] = v - p[3 * tp];          d[1] = v - p[3 * tp + 1];   d[2] = v - p[2 * tp + 2];   d[3] = v - p[tp + 3];
    d[4] = v - p[3];               d[5] = v - p[-tp + 3];      d[6] = v - p[-2 * tp + 2];  d[7] = v - p[-3 * tp + 1];
    d[8] = v - p[-3 * tp];         d[9] = v - p[-3 * tp - 1];  d[10] = v - p[-2 * tp - 2]; d[11] = v - p[-tp - 3];
    d[12] = v - p[-3];             d[13] = v - p[tp - 3];      d[14] = v - p[2 * tp - 2];  d[15] = v - p[3 * tp - 1];
    int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { mn8[k] = min(mn4[k], mn4[(k + 4) & 15]); mx8[k] = max(mx4[k], mx4[(k + 4) & 15]); }
    int best = -1000;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int mn9 = min(mn8[k], d[(k + 8) & 15]);
        int mx9 = max(mx8[k], d[(k + 8) & 15]);
        best = max(best, max(mn9, -mx9));
    }
    return best - 1;
}

// Four independent cells per 256-thread workgroup, one wave each (the tiles are ~1400 pixels: a whole
// workgroup per cell spends its life in barriers and the launch becomes dispatch bound: an EMPTY
// kernel over the 834k cell-workgroups of a 1024-frame batch already takes 5 ms).  Inside a wave the
// interior pixels are flattened in raster order, 64 per step, so all lanes work (interiors are ~31
// wide) and a ballot over a step is already in emission order -- no atomics or scans.  The tile keeps
// the 4-byte phase it has in global memory so loads and LDS stores are whole dwords.
template <int TILE>
#ifdef HVO_WPE_FAST
__attribute__((amdgpu_waves_per_eu(HVO_WPE_FAST)))
#endif
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ pyr0, size_t stride0, const uint8_t *__restrict__ lvl, size_t lvl_stride,
                                                    const LevelGeom *__restrict__ lev,
                                                    const CellDesc *__restrict__ cells, int ncells,
                                                    uint32_t *__restrict__ cell_kp, int *__restrict__ cell_cnt,
                                                    int iniTh, int minTh, int *__restrict__ flags)
{
    constexpr int TP = TILE + 8;                               // LDS pitch (multiple of 4; +3 phase bytes fit)
    constexpr int ND = TP / 4;
    constexpr int CAND_CAP = TILE * TILE / 4;
    constexpr int NSTEP = (TILE * TILE + 63) / 64;
    __shared__ __attribute__((aligned(16))) uint8_t tile_[4][TILE * TP];
    __shared__ __attribute__((aligned(16))) uint8_t sc_[4][TILE * TP];
    __shared__ unsigned long long sm_min[4][NSTEP], sm_ini[4][NSTEP];
    __shared__ unsigned short cand_[4][CAND_CAP];              // packed y << 8 | x
    __shared__ unsigned short pend_[4][128];                   // stage-1 survivors waiting for the rest of the pre-test
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int frame = blockIdx.y;
    int cell = blockIdx.x * 4 + wv;
    const bool live = cell < ncells;
    if (!live) cell = ncells - 1;                              // idle wave: run along harmlessly, write nothing
    uint8_t *tile = tile_[wv], *sc = sc_[wv];
    const CellDesc c = cells[cell];
    const LevelGeom L = lev[c.level];
    const uint8_t *img = c.level == 0 ? pyr0 + (size_t)frame * stride0 : lvl + (size_t)frame * lvl_stride + L.lvl_off;
    const int vw = c.vw, vh = c.vh;
    const int ax0 = c.x0 & ~3, sh = c.x0 - ax0;                // aligned start, byte phase
    {
        const int nd = (sh + vw + 3) >> 2;
        for (int i = lane; i < vh * ND; i += 64) {
            const int y = i / ND, dx = i - y * ND;
            uint32_t v = 0;
            if (dx < nd) v = *reinterpret_cast<const uint32_t *>(img + (size_t)(c.y0 + y) * L.pitch + ax0 + 4 * dx);
            reinterpret_cast<uint32_t *>(tile)[y * ND + dx] = v;
            reinterpret_cast<uint32_t *>(sc)[y * ND + dx] = 0;
        }
    }
    __syncthreads();
    const int iw = vw - 6, ih = vh - 6, P = iw * ih;
    const uint8_t *T0 = tile + sh;                             // T0[y*TP + x] = view pixel (x, y)
    uint8_t *S0 = sc + sh;
    // lane's first pixel and the per-step advance (64 = a*iw + b)
    const int adv_y = 64 / iw, adv_x = 64 - adv_y * iw;
    // ---- phase A: necessary conditions for every interior pixel, survivors compacted into an LDS list.
    // A 9-arc of the 16-ring contains ring pixel k or k+8 for every k, and all its pixels are on the
    // same side of the centre, so with cls = 1 (darker than v - t) | 2 (brighter than v + t):
    //   AND over the 8 opposite pairs of (cls[k] | cls[k+8]) != 0     (the test cv::FAST itself uses)
    // Stage 1 (the vertical pair 0 | 8) runs over the raster, 64 pixels a step; ~15 % of the pixels pass it but
    // nearly every step has a lane that does, so the other seven pairs are not chained behind it under
    // predication: survivors are queued (raster order) and the rest of the test runs on full waves of 64.
    int ncand = 0;
    {
        unsigned short *pend = pend_[wv];
        int npend = 0;
        auto rest = [&](int cnt) {                         // pairs 4|12, 2|10, 6|14, 1|9, 3|11, 5|13, 7|15 for pend[0, cnt)
            bool pass = false; int xy = 0;
            if (lane < cnt) {
                xy = pend[lane];
                const uint8_t *q = T0 + (xy >> 8) * TP + (xy & 0xFF);
                const int v = q[0], lo = v - minTh, hi = v + minTh;
#define CLS(o) ((q[o] < lo ? 1 : 0) | (q[o] > hi ? 2 : 0))
                int d = CLS(3 * TP) | CLS(-3 * TP);
                d &= CLS(3) | CLS(-3);
                d &= CLS(2 * TP + 2) | CLS(-2 * TP - 2);
                d &= CLS(-2 * TP + 2) | CLS(2 * TP - 2);
                d &= CLS(3 * TP + 1) | CLS(-3 * TP - 1);
                d &= CLS(TP + 3) | CLS(-TP - 3);
                d &= CLS(-TP + 3) | CLS(TP - 3);
                d &= CLS(-3 * TP + 1) | CLS(3 * TP - 1);
                pass = d != 0;
            }
            const unsigned long long m = __ballot(pass);
            if (pass) { const int p = ncand + __popcll(m & ((1ull << lane) - 1)); if (p < CAND_CAP) cand_[wv][p] = (unsigned short)xy; }
            ncand += __popcll(m);
        };
        int yy = lane / iw, xx = lane - yy * iw;
        for (int p0 = 0; p0 < P; p0 += 64) {
            bool pass = false;
            const int x = xx + 3, y = yy + 3;
            if (p0 + lane < P) {
                const uint8_t *q = T0 + y * TP + x;
                const int v = q[0], lo = v - minTh, hi = v + minTh;
                pass = (CLS(3 * TP) | CLS(-3 * TP)) != 0;                 // ring 0 | 8
#undef CLS
            }
            const unsigned long long m = __ballot(pass);
            if (pass) pend[npend + __popcll(m & ((1ull << lane) - 1))] = (unsigned short)((y << 8) | x);
            npend += __popcll(m);
            if (npend >= 64) {                                 // a full wave of survivors: finish their test
                rest(64);
                npend -= 64;
                const int t = lane < npend ? pend[64 + lane] : 0;
                if (lane < npend) pend[lane] = (unsigned short)t;
            }
            xx += adv_x; yy += adv_y; if (xx >= iw) { xx -= iw; yy++; }
        }
        if (npend > 0) rest(npend);
    }
    __syncthreads();
    // ---- phase B: exact score of the candidates, one per lane (~250 instructions each) ----
    if (ncand <= CAND_CAP) {
        for (int i = lane; i < ncand; i += 64) {
            const int xy = cand_[wv][i], x = xy & 0xFF, y = xy >> 8;
            const int s = fast_score(T0, TP, x, y, T0[y * TP + x]);
            S0[y * TP + x] = (uint8_t)(s >= minTh ? s : 0);
        }
    } else {                                                   // pathological tile: score everything
        for (int p = lane; p < P; p += 64) {
            const int y = p / iw + 3, x = p - (y - 3) * iw + 3;
            const int s = fast_score(T0, TP, x, y, T0[y * TP + x]);
            S0[y * TP + x] = (uint8_t)(s >= minTh ? s : 0);
        }
    }
    __syncthreads();
    // ---- strict 3x3 NMS and emission.  Only phase-A survivors can have a score, and their list is already in
    //      raster order, so both walk the candidate list (a step or two) instead of the whole interior; the
    //      full-raster walk remains for the pathological tile whose list overflowed. ----
    int n_ini = 0;
    uint32_t *out = cell_kp + ((size_t)frame * ncells + cell) * HVO_CELL_CAP;
    int pos = 0;
    if (ncand <= CAND_CAP) {
        int st = 0;
        for (int base = 0; base < ncand; base += 64, st++) {
            const int i = base + lane;
            bool ok = false; int v = 0;
            if (i < ncand) {
                const int xy = cand_[wv][i];
                const uint8_t *sp = S0 + (xy >> 8) * TP + (xy & 0xFF);
                v = sp[0];
                // neighbours outside the interior region are never written -> 0, like the zeroed score
                // rows/columns of the reference's per-view FAST call
                ok = v != 0 && v > sp[-1] && v > sp[1] && v > sp[-TP - 1] && v > sp[-TP] && v > sp[-TP + 1] &&
                     v > sp[TP - 1] && v > sp[TP] && v > sp[TP + 1];
            }
            const unsigned long long mm = __ballot(ok), mi = __ballot(ok && v >= iniTh);
            n_ini += __popcll(mi);
            if (lane == 0) { sm_min[wv][st] = mm; sm_ini[wv][st] = mi; }
        }
        __syncthreads();
        const bool use_ini = n_ini > 0;                        // iniThFAST survivors, else the minThFAST fallback
        st = 0;
        for (int base = 0; base < ncand; base += 64, st++) {
            const unsigned long long m = use_ini ? sm_ini[wv][st] : sm_min[wv][st];
            if (m) {
                if (live && ((m >> lane) & 1ull)) {
                    const int xy = cand_[wv][base + lane], x = xy & 0xFF, y = xy >> 8;
                    const int p = pos + __popcll(m & ((1ull << lane) - 1));
                    if (p < HVO_CELL_CAP) out[p] = (uint32_t)(x + c.ox) | ((uint32_t)(y + c.oy) << 12) | ((uint32_t)S0[y * TP + x] << 24);
                }
                pos += __popcll(m);
            }
        }
    } else {
        {
            int yy = lane / iw, xx = lane - yy * iw, st = 0;
            for (int p0 = 0; p0 < P; p0 += 64, st++) {
                bool ok = false; int v = 0;
                if (p0 + lane < P) {
                    const uint8_t *sp = S0 + (yy + 3) * TP + xx + 3;
                    v = sp[0];
                    ok = v != 0 && v > sp[-1] && v > sp[1] && v > sp[-TP - 1] && v > sp[-TP] && v > sp[-TP + 1] &&
                         v > sp[TP - 1] && v > sp[TP] && v > sp[TP + 1];
                }
                const unsigned long long mm = __ballot(ok), mi = __ballot(ok && v >= iniTh);
                n_ini += __popcll(mi);
                if (lane == 0) { sm_min[wv][st] = mm; sm_ini[wv][st] = mi; }
                xx += adv_x; yy += adv_y; if (xx >= iw) { xx -= iw; yy++; }
            }
        }
        __syncthreads();
        const bool use_ini = n_ini > 0;
        int yy = lane / iw, xx = lane - yy * iw, st = 0;
        for (int p0 = 0; p0 < P; p0 += 64, st++) {
            const unsigned long long m = use_ini ? sm_ini[wv][st] : sm_min[wv][st];
            if (m) {
                if (live && ((m >> lane) & 1ull)) {
                    const int x = xx + 3, y = yy + 3;
                    const int p = pos + __popcll(m & ((1ull << lane) - 1));
                    if (p < HVO_CELL_CAP) out[p] = (uint32_t)(x + c.ox) | ((uint32_t)(y + c.oy) << 12) | ((uint32_t)S0[y * TP + x] << 24);
                }
                pos += __popcll(m);
            }
            xx += adv_x; yy += adv_y; if (xx >= iw) { xx -= iw; yy++; }
        }
    }
    if (live && lane == 0) {
        if (pos > HVO_CELL_CAP) { atomicOr(&flags[frame], 1); pos = HVO_CELL_CAP; }
        cell_cnt[(size_t)frame * ncells + cell] = pos;
    }
}

// =====================================================================================
// K3: quadtree distribution -- one wave per (frame, level); the reference's serial semantics, evaluated a whole round at a time
// =====================================================================================
// DistributeOctTree walks a std::list: a round divides every node that holds more than one key (children pushed to the FRONT in order
// n1..n4, the node erased), and once the list is close to N nodes it divides one node at a time, largest first, until N is reached
// (ORBextractor.cc:588-735).  Within a round -- and within one pass of the largest-first loop -- the divisions do not depend on each
// other: what a division does to the list, the creation number of a child (the tie rule of the (size, node*) sort at 682: SURVEY H2)
// and the point at which the largest-first loop stops are prefix sums over the child counts of the nodes in processing order.  So:
//   * a LANE owns a node of the round (64 at a time): it counts its keys per quadrant, a wave scan numbers the children, finds the
//     stopping point and hands out slots, the lane writes its children's records and partitions its keys (stable) into the other of
//     two key buffers.  Nodes of more than 64 keys (the first two or three rounds) are counted and partitioned by the whole wave.
//   * the list is an ARRAY of slots, rebuilt per round: this round's children in reverse creation order, then the old list without
//     the divided nodes (a divided node's slot goes to its first child: `born[slot] == round` marks it).
// Per node in LDS: rec[slot] = {x0 | y0 << 16, x1 | y1 << 16, koff, nk | buffer << 31}; a node's keys ARE its candidates
// (x | y << 12 | score << 24) in kbuf[buffer][koff, koff + nk).  bNoMore <=> nk == 1.  At most max(N + 3, 4 nIni) nodes are alive
// (slot_cap, orb_build_plan).  Round 3's kernel divided one node at a time with the whole wave: ~2900 cycles a division, 6.85 ms per
// 8192 frames; its records were in global memory.
struct OctArgs {
    const LevelGeom *lev;
    const uint32_t *cell_kp; const int *cell_cnt; int ncells;
    uint32_t *cand; int *keys; int *keys_tmp;
    uint32_t *lvl_kp; int *lvl_cnt; int *flags;
    int cand_total, kp_total, nlevels, slot_cap;
};
#define OCT_SLOT_BYTES 40                              // uint4 record, an entry in each of the two round lists, born + two list arrays + creation order (u16)

static __device__ __forceinline__ int cand_x(uint32_t c) { return c & 0xFFF; }
static __device__ __forceinline__ int cand_y(uint32_t c) { return (c >> 12) & 0xFFF; }
static __device__ __forceinline__ int cand_s(uint32_t c) { return c >> 24; }

extern __shared__ uint4 oct_lds[];

__global__ __launch_bounds__(64) void k_octree(OctArgs a)
{
    const int level = blockIdx.x, frame = blockIdx.y, lane = threadIdx.x;
    const LevelGeom L = a.lev[level];
    uint32_t *cand = a.cand + (size_t)frame * a.cand_total + L.cand_off;
    int *kbuf0 = a.keys + (size_t)frame * a.cand_total + L.cand_off;
    int *kbuf1 = a.keys_tmp + (size_t)frame * a.cand_total + L.cand_off;
    uint32_t *outkp = a.lvl_kp + (size_t)frame * a.kp_total + L.kp_off;
    int *outcnt = a.lvl_cnt + (size_t)frame * a.nlevels + level;
    const unsigned long long lt = lanemask_lt();
    const int cap = a.slot_cap;
    uint4 *rec = oct_lds;
    int2 *DA = reinterpret_cast<int2 *>(rec + cap), *DB = DA + cap;              // the nodes a round divides: (keys, creation number | slot << 16)
    unsigned short *born = reinterpret_cast<unsigned short *>(DB + cap);          // the round a slot's record was written in
    unsigned short *lstO = born + cap, *lstN = lstO + cap, *tmpC = lstN + cap;    // the list, the list being built, this round's children in creation order

    // ---- gather the cells' candidates in cell order (vToDistributeKeys) ----
    // 64 cells at a time: their counts in one load, a wave prefix sum for the offsets (LDS), then the chunk's candidates copied with all
    // lanes (a binary search over the 65 offsets finds an element's cell) -- not one dependent count load per cell (815 per frame)
    __shared__ int goff[65];
    int nc = 0;
    bool overflow = false;
    const int N = L.nfeat;
    const int bw = L.maxBX - L.minBX, bh = L.maxBY - L.minBY;
    const int nIni = (int)roundf(__fdiv_rn((float)bw, (float)bh));
    // one initial node (any image that is not wider than 1.5 x its height): it holds every candidate, in this order -- the gather writes its keys
    uint32_t *gdst = nIni == 1 ? reinterpret_cast<uint32_t *>(kbuf0) : cand;
    for (int cbase = 0; cbase < L.ncells; cbase += 64) {
        const int ci = cbase + lane;
        const size_t cidx0 = (size_t)frame * a.ncells + L.cell_off + cbase;
        int cnt = ci < L.ncells ? a.cell_cnt[cidx0 + lane] : 0;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        goff[lane + 1] = incl;
        if (lane == 0) goff[0] = 0;
        __syncthreads();
        int total = goff[64];
        if (nc + total > L.cand_cap) { total = L.cand_cap - nc; overflow = true; }
        for (int t = lane; t < total; t += 64) {
            int lo = 0, hi = 64;                       // the cell c with goff[c] <= t < goff[c + 1]
#pragma unroll
            for (int step = 0; step < 6; step++) { const int mid = (lo + hi) >> 1; if (goff[mid] <= t) lo = mid; else hi = mid; }
            gdst[nc + t] = a.cell_kp[(cidx0 + lo) * HVO_CELL_CAP + (t - goff[lo])];
        }
        nc += total;
        __syncthreads();
    }
    if (overflow && lane == 0) atomicOr(&a.flags[frame], 2);
    if (nc == 0 || L.nCols < 1 || L.nRows < 1) { if (lane == 0) *outcnt = 0; return; }
    __syncthreads();

    if (nIni < 1) { if (lane == 0) *outcnt = 0; return; }
    const float hX = __fdiv_rn((float)bw, (float)nIni);

    int nslot = 0;            // slots handed out
    int nseq = 0;             // nodes created
    int nlist = 0;            // nodes in the list (lNodes.size())
    int nexp0 = 0;            // initial nodes that hold more than one key

    // ---- initial nodes: stable filter of candidates by (int)(x / hX) ----
    {
        int koff = 0;
        for (int i = 0; i < nIni; i++) {
            int cnt = 0;
            if (nIni == 1) cnt = nc;                   // (int)(x / hX) == 0 for every x < bw: the gather has written this node's keys
            else for (int b = 0; b < nc; b += 64) {
                int k = b + lane;
                bool in = false;
                if (k < nc) in = ((int)__fdiv_rn((float)cand_x(cand[k]), hX)) == i;
                unsigned long long m = __ballot(in);
                if (in) kbuf0[koff + cnt + __popcll(m & lt)] = (int)cand[k];
                cnt += __popcll(m);
            }
            if (cnt > 0) {   // empty initial nodes are erased right away (ORBextractor.cc:580-581)
                if (nslot >= cap) { if (lane == 0) { atomicOr(&a.flags[frame], 4); *outcnt = 0; } return; }
                if (lane == 0) {
                    rec[nslot] = make_uint4((unsigned)(int)(hX * (float)i), (unsigned)(int)(hX * (float)(i + 1)) | ((unsigned)bh << 16), (unsigned)koff, (unsigned)cnt);
                    born[nslot] = 0;
                    lstO[nlist] = (unsigned short)nslot;
                    if (cnt > 1) DA[nexp0] = make_int2(cnt, nseq | (nslot << 16));           // the first round's nodes to divide, in list order
                }
                if (cnt > 1) nexp0++;
                nslot++; nseq++; nlist++;
            }
            koff += cnt;
        }
    }
    __syncthreads();                                   // (the keys just written are read below: drains the stores)

    int round = 0;
    bool nodes_full = false;
    // One round: the nodes D[0, nD) (backwards if `rev`) are divided in that order -- with `cut`, only until the list holds N nodes.
    // The children that hold more than one key go to D2 in creation order; returns how many.
    auto run_round = [&](const int2 *D, int nD, bool rev, bool cut, int2 *D2) -> int {
        round++;
        int stepC = 0, stepG = 0, grown = 0;           // children / children with more than one key / growth of the list, so far in this round
        bool stop = false;
        for (int base = 0; base < nD && !stop; base += 64) {
            const int i = base + lane;
            const bool valid = i < nD;
            const int ey = valid ? D[rev ? nD - 1 - i : i].y : 0;
            const int slot = (int)((unsigned)ey >> 16);
            const uint4 R = valid ? rec[slot] : make_uint4(0, 0, 0, 0);
            const int x0 = (int)(R.x & 0xFFFFu), y0 = (int)(R.x >> 16), x1 = (int)(R.y & 0xFFFFu), y1 = (int)(R.y >> 16);
            const int koff = (int)R.z, nk = (int)(R.w & 0x7FFFFFFFu), buf = (int)(R.w >> 31);
            const int mx = x0 + (int)ceilf((float)(x1 - x0) * 0.5f), my = y0 + (int)ceilf((float)(y1 - y0) * 0.5f);
            const int *src = buf ? kbuf1 : kbuf0;
            int *dst = buf ? kbuf0 : kbuf1;
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            // ---- keys per quadrant: nodes of more than 64 keys by the whole wave, one after the other ...
            const unsigned long long bigm = __ballot(valid && nk > 64);
            for (unsigned long long bm = bigm; bm; bm &= bm - 1) {
                const int o = __ffsll(bm) - 1;
                const int bk = __shfl(koff, o), bn = __shfl(nk, o), bmx = __shfl(mx, o), bmy = __shfl(my, o);
                const int *bs = __shfl(buf, o) ? kbuf1 : kbuf0;
                int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                for (int b = 0; b < bn; b += 256) {
                    uint32_t cd[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int k = b + 64 * j + lane; cd[j] = k < bn ? (uint32_t)bs[bk + k] : 0u; }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int cls = b + 64 * j + lane < bn ? (cand_x(cd[j]) < bmx ? 0 : 1) + (cand_y(cd[j]) < bmy ? 0 : 2) : -1;
                        t0 += __popcll(__ballot(cls == 0)); t1 += __popcll(__ballot(cls == 1)); t2 += __popcll(__ballot(cls == 2)); t3 += __popcll(__ballot(cls == 3));
                    }
                }
                if (lane == o) { c0 = t0; c1 = t1; c2 = t2; c3 = t3; }
            }
            // ---- ... the others each by its lane
            {
                const int mynk = (valid && nk <= 64) ? nk : 0;
                for (int k = 0; __ballot(k < mynk) != 0ull; k += 2) {
                    const uint32_t ca = k < mynk ? (uint32_t)src[koff + k] : 0u, cb = k + 1 < mynk ? (uint32_t)src[koff + k + 1] : 0u;
                    if (k < mynk) { const int cls = (cand_x(ca) < mx ? 0 : 1) + (cand_y(ca) < my ? 0 : 2); c0 += cls == 0; c1 += cls == 1; c2 += cls == 2; c3 += cls == 3; }
                    if (k + 1 < mynk) { const int cls = (cand_x(cb) < mx ? 0 : 1) + (cand_y(cb) < my ? 0 : 2); c0 += cls == 0; c1 += cls == 1; c2 += cls == 2; c3 += cls == 3; }
                }
            }
            // ---- number the children: one scan of (children | list growth << 10 | children with more than one key << 20)
            const int nch = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);               // >= 1: only nodes of two or more keys are divided
            const int packed = valid ? (nch | ((nch - 1) << 10) | (((c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1)) << 20)) : 0;
            int incl = packed;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            bool ex = valid;
            if (cut) {                                  // "if ((int)lNodes.size() >= N) break" after each division (ORBextractor.cc:725)
                const unsigned long long cm = __ballot(valid && nlist + grown + ((incl >> 10) & 0x3FF) >= N);
                if (cm) { ex = valid && lane <= __ffsll(cm) - 1; stop = true; }
            }
            const unsigned long long exm = __ballot(ex);
            const int tot = __shfl(incl, 63 - __clzll(exm));                          // (exm != 0: lane 0 is valid)
            const int totC = tot & 0x3FF, totI = (tot >> 10) & 0x3FF, totG = tot >> 20;
            const int excl = incl - packed;
            const int Cpre = excl & 0x3FF, Ipre = (excl >> 10) & 0x3FF, Gpre = excl >> 20;
            if (nslot + totI > cap || nseq + stepC + totC > 0xFFFF) { nodes_full = true; return 0; }
            const int o1 = c0, o2 = c0 + c1, o3 = c0 + c1 + c2;
            if (ex) {
                int r = 0, rr = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int cq = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
                    if (cq > 0) {
                        const int cslot = r == 0 ? slot : nslot + Ipre + r - 1;      // the first child takes the divided node's slot
                        const int cidx = stepC + Cpre + r;
                        const int cx0 = (q & 1) ? mx : x0, cx1 = (q & 1) ? x1 : mx, cy0 = (q & 2) ? my : y0, cy1 = (q & 2) ? y1 : my;
                        const int ko = koff + (q == 0 ? 0 : q == 1 ? o1 : q == 2 ? o2 : o3);
                        rec[cslot] = make_uint4((unsigned)cx0 | ((unsigned)cy0 << 16), (unsigned)cx1 | ((unsigned)cy1 << 16), (unsigned)ko, (unsigned)cq | ((unsigned)(buf ^ 1) << 31));
                        born[cslot] = (unsigned short)round;
                        tmpC[cidx] = (unsigned short)cslot;
                        if (cq > 1) { D2[stepG + Gpre + rr] = make_int2(cq, (nseq + cidx) | (cslot << 16)); rr++; }
                        r++;
                    }
                }
            }
            // ---- the keys, stable, into the other buffer: the lanes' own nodes ...
            {
                const int mynk = (ex && nk <= 64) ? nk : 0;
                int d0 = koff, d1 = koff + o1, d2 = koff + o2, d3 = koff + o3;
                for (int k = 0; __ballot(k < mynk) != 0ull; k += 2) {
                    const uint32_t ca = k < mynk ? (uint32_t)src[koff + k] : 0u, cb = k + 1 < mynk ? (uint32_t)src[koff + k + 1] : 0u;
                    if (k < mynk) {
                        const int cls = (cand_x(ca) < mx ? 0 : 1) + (cand_y(ca) < my ? 0 : 2);
                        const int d = cls == 0 ? d0 : cls == 1 ? d1 : cls == 2 ? d2 : d3;
                        dst[d] = (int)ca; d0 += cls == 0; d1 += cls == 1; d2 += cls == 2; d3 += cls == 3;
                    }
                    if (k + 1 < mynk) {
                        const int cls = (cand_x(cb) < mx ? 0 : 1) + (cand_y(cb) < my ? 0 : 2);
                        const int d = cls == 0 ? d0 : cls == 1 ? d1 : cls == 2 ? d2 : d3;
                        dst[d] = (int)cb; d0 += cls == 0; d1 += cls == 1; d2 += cls == 2; d3 += cls == 3;
                    }
                }
            }
            // ---- ... and the big ones by the whole wave
            for (unsigned long long bm = bigm & exm; bm; bm &= bm - 1) {
                const int o = __ffsll(bm) - 1;
                const int bk = __shfl(koff, o), bn = __shfl(nk, o), bmx = __shfl(mx, o), bmy = __shfl(my, o), bb = __shfl(buf, o);
                const int bo1 = __shfl(o1, o), bo2 = __shfl(o2, o), bo3 = __shfl(o3, o);
                const int *bs = bb ? kbuf1 : kbuf0;
                int *bd = bb ? kbuf0 : kbuf1;
                int r0 = bk, r1 = bk + bo1, r2 = bk + bo2, r3 = bk + bo3;
                for (int b = 0; b < bn; b += 256) {
                    uint32_t cd[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int k = b + 64 * j + lane; cd[j] = k < bn ? (uint32_t)bs[bk + k] : 0u; }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int cls = b + 64 * j + lane < bn ? (cand_x(cd[j]) < bmx ? 0 : 1) + (cand_y(cd[j]) < bmy ? 0 : 2) : -1;
                        const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
                        if (cls >= 0) {
                            const int d = cls == 0 ? r0 + __popcll(m0 & lt) : cls == 1 ? r1 + __popcll(m1 & lt) : cls == 2 ? r2 + __popcll(m2 & lt) : r3 + __popcll(m3 & lt);
                            bd[d] = (int)cd[j];
                        }
                        r0 += __popcll(m0); r1 += __popcll(m1); r2 += __popcll(m2); r3 += __popcll(m3);
                    }
                }
            }
            stepC += totC; stepG += totG; nslot += totI; grown += totI;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the list: this round's children, last created first, then the old list without the divided nodes ----
        for (int j = lane; j < stepC; j += 64) lstN[j] = tmpC[stepC - 1 - j];
        int pos = stepC;
        for (int base = 0; base < nlist; base += 64) {
            const int p = base + lane;
            const int s = p < nlist ? lstO[p] : 0;
            const bool keep = p < nlist && born[s] != (unsigned short)round;
            const unsigned long long m = __ballot(keep);
            if (keep) lstN[pos + __popcll(m & lt)] = (unsigned short)s;
            pos += __popcll(m);
        }
        nlist = pos;
        { unsigned short *t = lstO; lstO = lstN; lstN = t; }
        nseq += stepC;
        __syncthreads();                               // (keys written in this round are read in the next: drains the stores)
        return stepG;
    };

    // A round divides every node that holds more than one key, in LIST order.  Those are exactly the children of the previous round that
    // got more than one key -- pushed to the front one after the other, so the list holds them in the reverse of their creation order --
    // (the initial nodes, in creation order, for the first round).
    bool finish = false;
    int2 *Dcur = DA, *Dnxt = DB;
    int nD = nexp0; bool rev = false;
    while (!finish) {
        int prevSize = nlist;
        const int nToExpand = run_round(Dcur, nD, rev, false, Dnxt);
        { int2 *t = Dcur; Dcur = Dnxt; Dnxt = t; } nD = nToExpand; rev = true;
        if (nodes_full) break;
        if (nlist >= N || nlist == prevSize) {
            finish = true;
        } else if (nlist + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = nlist;
                // (keys, creation number) DESCENDING into the free array: the order the reference pops vPrevSizeAndPointerToNode in; pairs are unique
                for (int i = lane; i < nD; i += 64) {
                    const int2 me = Dcur[i];
                    int r = 0;
                    for (int j = 0; j < nD; j++) { const int2 o = Dcur[j]; r += (o.x > me.x) || (o.x == me.x && (o.y & 0xFFFF) > (me.y & 0xFFFF)); }
                    Dnxt[r] = me;
                }
                __builtin_amdgcn_wave_barrier();
                nD = run_round(Dnxt, nD, false, true, Dcur);          // (the unsorted array is free: the round lists its children there)
                if (nodes_full) break;
                if (nlist >= N || nlist == prevSize) finish = true;
            }
        }
    }
    if (nodes_full) { if (lane == 0) { atomicOr(&a.flags[frame], 4); *outcnt = 0; } return; }

    // ---- retain the best point of each node, list order (ORBextractor.cc:737-758) ----
    const int cntn = min(nlist, L.kp_cap);
    for (int i = lane; i < cntn; i += 64) {
        const uint4 R = rec[lstO[i]];
        const int *src = (R.w >> 31) ? kbuf1 : kbuf0;
        const int nk = (int)(R.w & 0x7FFFFFFFu);
        uint32_t best = (uint32_t)src[R.z];
        for (int k = 1; k < nk; k++) { uint32_t c = (uint32_t)src[R.z + k]; if (cand_s(c) > cand_s(best)) best = c; }
        outkp[i] = best;
    }
    if (lane == 0) *outcnt = cntn;
}

// =====================================================================================
// K5a: 7x7 Gaussian blur, sigma 2, u8 -> u8 (OpenCV 3.2 fixed-point path)
// =====================================================================================
// Row pass with integer taps {18,34,49,55,49,34,18} (sum 257) fits u16; column pass in int32,
// then: columns x < (w & ~3) round-half-to-even of s/65536 (the SSE2 float path of
// SymmColumnVec_32s8u -- exact in float here), the last w%4 columns (s + 32768) >> 16.
// Register sliding-window formulation (no LDS, no barriers): a thread owns a 4-pixel-wide column strip
// of BLUR_TH rows.  Per input row it loads the 12 bytes [x0-4, x0+8) as three aligned dwords, forms the
// byte windows with v_alignbyte and evaluates the 7-tap row sum of each of its 4 pixels with two
// v_dot4_u32_u8; the last 7 row sums per pixel stay in registers for the column pass.  Output is one
// dword store per row.  Strips that touch the left/right image border load from clamped addresses and permute
// the reflected bytes into place (EdgeSel, hvo_internal.hpp): a byte-wise border path would be issued for every
// wave that contains a border strip, two tile columns in three at 640 pixels.
#define BLUR_TW 256            // pixels per workgroup row (64 threads x 4 px)
#define BLUR_TH 32             // rows per wave; a workgroup (4 waves) covers 128 rows
struct BlurWin { unsigned W0, W1, W2; };
static __device__ __forceinline__ BlurWin blur_load(const uint8_t *__restrict__ row, int o0, int x0, int o2, const EdgeSel &es)
{
    BlurWin o;
    o.W0 = *reinterpret_cast<const uint32_t *>(row + o0);      // x0 - 4, clamped into the row for the leftmost strip
    o.W1 = *reinterpret_cast<const uint32_t *>(row + x0);
    o.W2 = *reinterpret_cast<const uint32_t *>(row + o2);      // x0 + 4, clamped into the row's pitch
    edge_fix(es, o.W0, o.W1, o.W2);                            // REFLECT_101 at the left / right image border (identity elsewhere)
    return o;
}
static __device__ __forceinline__ void blur_row_sums(const BlurWin &W, unsigned klo, unsigned khi, int hs[4])
{
    // pixel j reads window bytes j+1 .. j+7
    const unsigned A0 = __builtin_amdgcn_alignbyte(W.W1, W.W0, 1), B0 = __builtin_amdgcn_alignbyte(W.W2, W.W1, 1);
    const unsigned A1 = __builtin_amdgcn_alignbyte(W.W1, W.W0, 2), B1 = __builtin_amdgcn_alignbyte(W.W2, W.W1, 2);
    const unsigned A2 = __builtin_amdgcn_alignbyte(W.W1, W.W0, 3), B2 = __builtin_amdgcn_alignbyte(W.W2, W.W1, 3);
    hs[0] = (int)__builtin_amdgcn_udot4(A0, klo, __builtin_amdgcn_udot4(B0, khi, 0u, false), false);
    hs[1] = (int)__builtin_amdgcn_udot4(A1, klo, __builtin_amdgcn_udot4(B1, khi, 0u, false), false);
    hs[2] = (int)__builtin_amdgcn_udot4(A2, klo, __builtin_amdgcn_udot4(B2, khi, 0u, false), false);
    hs[3] = (int)__builtin_amdgcn_udot4(W.W1, klo, __builtin_amdgcn_udot4(W.W2, khi, 0u, false), false);
}

#ifdef HVO_WPE_BLUR7
__attribute__((amdgpu_waves_per_eu(HVO_WPE_BLUR7)))
#endif
__global__ __launch_bounds__(256) void k_blur7(const uint8_t *__restrict__ pyr0, size_t stride0, const uint8_t *__restrict__ lvl, size_t lvl_stride,
                                               uint8_t *__restrict__ blur, size_t blur_stride, const LevelGeom *__restrict__ lev, const int4 *__restrict__ tiles,
                                               int k0, int k1, int k2, int k3)
{
    const int4 t = tiles[blockIdx.x];
    const LevelGeom L = lev[t.x];
    const uint8_t *src = t.x == 0 ? pyr0 + (size_t)blockIdx.y * stride0 : lvl + (size_t)blockIdx.y * lvl_stride + L.lvl_off;
    uint8_t *dst = blur + (size_t)blockIdx.y * blur_stride + L.img_off;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = t.y * BLUR_TW + lane * 4;
    const int y0 = (t.z * 4 + wv) * BLUR_TH;
    if (x0 >= L.w || y0 >= L.h) return;
    const EdgeSel es = edge_sel(x0, L.w);
    const int o0 = max(x0 - 4, 0), o2 = min(x0 + 4, L.pitch - 4);
    const unsigned klo = (unsigned)k0 | ((unsigned)k1 << 8) | ((unsigned)k2 << 16) | ((unsigned)k3 << 24);
    const unsigned khi = (unsigned)k2 | ((unsigned)k1 << 8) | ((unsigned)k0 << 16);
    const int wv4 = L.w & ~3;
    const int rows = min(BLUR_TH, L.h - y0);
#define BLUR_ROW(i) (src + (size_t)reflect101(y0 + (i) - 3, L.h) * L.pitch)
    // all six priming rows are fetched before any is consumed; inside the loop the window of row r+2 is
    // requested while row r is being processed (two loads in flight per thread)
    BlurWin pw[6];
#pragma unroll
    for (int r = 0; r < 6; r++) pw[r] = blur_load(BLUR_ROW(r), o0, x0, o2, es);
    BlurWin n0 = blur_load(BLUR_ROW(6), o0, x0, o2, es);
    BlurWin n1 = blur_load(BLUR_ROW(7), o0, x0, o2, es);
    int h[7][4];
#pragma unroll
    for (int r = 0; r < 6; r++) blur_row_sums(pw[r], klo, khi, h[r]);
    for (int r = 0; r < rows; r++) {
        const BlurWin cur = n0;
        n0 = n1;
        n1 = blur_load(BLUR_ROW(r + 8), o0, x0, o2, es);       // rows past the band are loaded (reflected) but unused
        blur_row_sums(cur, klo, khi, h[6]);
        unsigned out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sv = k0 * (h[0][j] + h[6][j]) + k1 * (h[1][j] + h[5][j]) + k2 * (h[2][j] + h[4][j]) + k3 * h[3][j];
            int q;
            if (x0 + j < wv4) { q = sv >> 16; const int rem = sv & 0xFFFF; if (rem > 32768 || (rem == 32768 && (q & 1))) q++; }
            else q = (sv + 32768) >> 16;
            q = min(q, 255);
            if (x0 + j < L.w) out |= (unsigned)q << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)(y0 + r) * L.pitch + x0) = out;   // pitch % 64 == 0: stays in the row
#pragma unroll
        for (int q = 0; q < 6; q++) {
#pragma unroll
            for (int j = 0; j < 4; j++) h[q][j] = h[q + 1][j];
        }
    }
#undef BLUR_ROW
}

// =====================================================================================
// host side
// =====================================================================================
static int round_half_even_f(float v) { return (int)lrintf(v); }
static int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

int orb_init_tables(hvo_ctx *ctx)
{
    const hvo_params &p = ctx->p;
    if (p.orb_nlevels < 1 || p.orb_nlevels > HVO_MAX_LEVELS || p.orb_nfeatures < 1 || !(p.orb_scale_factor > 1.0f))
        return HVO_ERR_INVALID_ARG;
    // ORBextractor.cc:413-444
    ctx->scale[0] = 1.0f;
    for (int i = 1; i < p.orb_nlevels; i++) ctx->scale[i] = ctx->scale[i - 1] * p.orb_scale_factor;
    for (int i = 0; i < p.orb_nlevels; i++) ctx->inv_scale[i] = 1.0f / ctx->scale[i];
    float factor = 1.0f / p.orb_scale_factor;
    float nd = p.orb_nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p.orb_nlevels));
    int sum = 0;
    for (int l = 0; l < p.orb_nlevels - 1; l++) { ctx->nfeat[l] = round_half_even_f(nd); sum += ctx->nfeat[l]; nd *= factor; }
    ctx->nfeat[p.orb_nlevels - 1] = std::max(p.orb_nfeatures - sum, 0);
    // ORBextractor.cc:452-467 (umax of the circular patch)
    int *umax = ctx->umax;
    int v, v0, vmax = (int)floor(15 * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(15 * sqrtf(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(sqrt(225.0 - v * v));
    for (v = 15, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
    static const int8_t pattern[1024] = {
#include "orb_pattern.inc"
    };
    HVO_HIP(hipMalloc(&ctx->d_pattern, 1024));
    HVO_HIP(hipMemcpy(ctx->d_pattern, pattern, 1024, hipMemcpyHostToDevice));
    HVO_HIP(hipMalloc(&ctx->d_umax, 16 * sizeof(int)));
    HVO_HIP(hipMemcpy(ctx->d_umax, umax, 16 * sizeof(int), hipMemcpyHostToDevice));
    HVO_HIP(hipDeviceSynchronize());     // null-stream copies are not ordered against the non-blocking ctx stream
    return HVO_OK;
}

void orb_free_plan(hvo_ctx *ctx)
{
    OrbPlan &P = ctx->orb;
    void *ptrs[] = { P.d_lev, P.d_cells, P.d_rs_xofs, P.d_rs_xalpha, P.d_rs_yofs, P.d_rs_ybeta, P.d_tiles, P.d_pyr_base, P.d_blur,
                     P.d_cell_kp, P.d_cell_cnt, P.d_cand, P.d_keys, P.d_keys_tmp,
                     P.d_lvl_kp, P.d_lvl_cnt, P.d_kp, P.d_desc, P.d_nkp, P.d_flags, P.d_ltiles, P.d_kpchunks, P.d_lvl_base };
    for (void *q : ptrs) if (q) (void)hipFree(q);
    P = OrbPlan();
}

template <class T> static int dev_alloc(hvo_ctx *ctx, T **p, size_t n)
{
    HVO_HIP(hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return HVO_OK;
}
template <class T> static int dev_upload(hvo_ctx *ctx, T **p, const std::vector<T> &v)
{
    int rc = dev_alloc(ctx, p, v.size());
    if (rc) return rc;
    if (!v.empty()) HVO_HIP(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return HVO_OK;
}

// GaussianBlur(7 x 7, sigma 2) of every level of the chunk's frames in the float-kernel reading (HVO_READING_BLUR_FLOAT, readings.hip)
int orb_blur_float_run(hvo_ctx *ctx, int c0, int m, hipStream_t st)
{
    OrbPlan &P = ctx->orb;
    for (int l = 0; l < P.nlevels; l++) {
        const LevelGeom &L = P.lev[l];
        const uint8_t *src = l == 0 ? P.d_pyr + (size_t)c0 * P.pyr_bytes : P.d_lvl + L.lvl_off;
        const int rc = readings_gblur_enqueue(st, src, l == 0 ? P.pyr_bytes : P.lvl_bytes, L.pitch, L.w, L.h, P.d_blur + L.img_off, P.blur_bytes, L.pitch, m, 7, 2.0, true);
        if (rc) return rc;
    }
    return HVO_OK;
}

static int orb_build_plan(hvo_ctx *ctx, int w, int h, int batch);
// The cache key (w, h, batch) stays valid only if every table and slab behind it exists: a geometry that is rejected
// half-way (a level below 38 pixels, a cell larger than the LDS tile, an allocation failure) frees the partial plan,
// which resets the key, so that the next call with the same geometry is rejected again instead of running on null slabs.
bool orb_plan_covers(const hvo_ctx *ctx, int w, int h, int batch) { const OrbPlan &P = ctx->orb; return P.w == w && P.h == h && P.batch >= batch; }
int orb_ensure_plan(hvo_ctx *ctx, int w, int h, int batch)
{
    OrbPlan &P = ctx->orb;
    if (P.w == w && P.h == h && P.batch >= batch) return HVO_OK;
    if (w < 64 || h < 64 || w > 4095 || h > 4095) return HVO_ERR_UNSUPPORTED;
    orb_free_plan(ctx);
    const int rc = orb_build_plan(ctx, w, h, batch);
    if (rc) orb_free_plan(ctx);
    return rc;
}

static int orb_build_plan(hvo_ctx *ctx, int w, int h, int batch)
{
    OrbPlan &P = ctx->orb;
    const int nl = ctx->p.orb_nlevels;
    P.w = w; P.h = h; P.nlevels = nl; P.batch = batch; P.oct_slot_cap = 0;
    std::vector<CellDesc> cells;
    std::vector<int> xofs, xalpha, yofs, ybeta;
    std::vector<int4> tiles;
    size_t off = 0;
    int cand_total = 0, node_total = 0, kp_total = 0;
    for (int l = 0; l < nl; l++) {
        LevelGeom &L = P.lev[l];
        memset(&L, 0, sizeof(L));
        L.w = round_half_even_f((float)w * ctx->inv_scale[l]);      // ORBextractor.cc:1110
        L.h = round_half_even_f((float)h * ctx->inv_scale[l]);
        if (L.w < 38 || L.h < 38) return HVO_ERR_UNSUPPORTED;
        L.pitch = (L.w + 63) & ~63;
        L.img_off = off;
        L.lvl_off = l >= 1 ? off - P.lev[0].img_off - (size_t)P.lev[0].pitch * P.lev[0].h : 0;
        off += (size_t)L.pitch * L.h;
        L.scale = ctx->scale[l];
        L.scaled_patch = (int)(31 * ctx->scale[l]);
        L.nfeat = ctx->nfeat[l];
        // ORBextractor.cc:769-785
        L.minBX = L.minBY = HVO_EDGE_THRESHOLD - 3;
        L.maxBX = L.w - HVO_EDGE_THRESHOLD + 3; L.maxBY = L.h - HVO_EDGE_THRESHOLD + 3;
        const float width = (float)(L.maxBX - L.minBX), height = (float)(L.maxBY - L.minBY);
        L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);
        L.cell_off = (int)cells.size();
        if (L.nCols >= 1 && L.nRows >= 1) {
            L.wCell = (int)ceilf(width / L.nCols); L.hCell = (int)ceilf(height / L.nRows);
            for (int i = 0; i < L.nRows; i++) {
                const float iniY = (float)(L.minBY + i * L.hCell);
                float maxY = iniY + L.hCell + 6;
                if (iniY >= L.maxBY - 3) continue;
                if (maxY > L.maxBY) maxY = (float)L.maxBY;
                for (int j = 0; j < L.nCols; j++) {
                    const float iniX = (float)(L.minBX + j * L.wCell);
                    float maxX = iniX + L.wCell + 6;
                    if (iniX >= L.maxBX - 6) continue;
                    if (maxX > L.maxBX) maxX = (float)L.maxBX;
                    CellDesc c;
                    c.level = (short)l; c.x0 = (short)iniX; c.y0 = (short)iniY;
                    c.vw = (short)((int)maxX - (int)iniX); c.vh = (short)((int)maxY - (int)iniY);
                    c.ox = (short)(j * L.wCell); c.oy = (short)(i * L.hCell); c.pad = 0;
                    if (c.vw > HVO_CELL_TILE - 3 || c.vh > HVO_CELL_TILE) return HVO_ERR_UNSUPPORTED;
                    P.max_cell = std::max(P.max_cell, (int)std::max(c.vw, c.vh));
                    if (c.vw < 7 || c.vh < 7) continue;      // cv::FAST finds nothing in such a view
                    cells.push_back(c);
                }
            }
        }
        L.ncells = (int)cells.size() - L.cell_off;
        L.cand_off = cand_total;
        // strict 3x3 NMS leaves at most one survivor per 2x2 block: <= 256 per (<=32x32) cell,
        // so the per-cell slab capacity is also the exact bound for the level's candidate list
        L.cand_cap = L.ncells * HVO_CELL_CAP;
        cand_total += L.cand_cap;
        L.node_off = 0; L.node_cap = 6 * L.nfeat + 256;       // (nodes a level may CREATE: k_octree's creation numbers are 16 bits wide)
        if (L.node_cap > 0xFFFF) return HVO_ERR_UNSUPPORTED;
        {   // nodes alive at once in k_octree: the list stops at nfeat (+ 3), the unconditional first round makes up to 4 nIni
            const int bw = L.maxBX - L.minBX, bh = L.maxBY - L.minBY;
            const int nIni = bh > 0 ? (int)roundf((float)bw / (float)bh) : 0;
            P.oct_slot_cap = std::max(P.oct_slot_cap, (std::max(L.nfeat + 3, 4 * nIni) + 5 + 3) & ~3);
        }
        L.kp_off = kp_total;
        L.kp_cap = L.nfeat + 8;
        kp_total += L.kp_cap;
        // resize tables for level l (from l-1): cv::resize INTER_LINEAR fixed-point coefficients
        L.rs_off = (int)xofs.size(); L.ry_off = (int)yofs.size();
        if (l > 0) {
            const LevelGeom &S = P.lev[l - 1];
            const double scale_x = 1. / ((double)L.w / S.w), scale_y = 1. / ((double)L.h / S.h);
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_f(fx); fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
                short a0 = sat_short(round_half_even_f((1.f - fx) * 2048)), a1 = sat_short(round_half_even_f(fx * 2048));
                xofs.push_back(sx);
                xalpha.push_back((int)((uint32_t)(uint16_t)a0 | ((uint32_t)(uint16_t)a1 << 16)));
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor_f(fy); fy -= sy;
                short b0 = sat_short(round_half_even_f((1.f - fy) * 2048)), b1 = sat_short(round_half_even_f(fy * 2048));
                int sy0 = std::min(std::max(sy, 0), S.h - 1), sy1 = std::min(std::max(sy + 1, 0), S.h - 1);
                yofs.push_back(sy0 | (sy1 << 16));
                ybeta.push_back((int)((uint32_t)(uint16_t)b0 | ((uint32_t)(uint16_t)b1 << 16)));
            }
        }
        // k_resize_dw needs every thread's four left taps within 8 bytes of its aligned start (scale factor <= 4/3)
        P.resize_dw[l] = false;
        if (l > 0 && !getenv("HVO_RESIZE_BYTES")) {
            bool ok = true;
            const int *xo = xofs.data() + L.rs_off;
            for (int dx0 = 0; dx0 < L.w && ok; dx0 += 4) ok = xo[std::min(dx0 + 3, L.w - 1)] - (xo[dx0] & ~3) <= 7;
            P.resize_dw[l] = ok;
        }
        L.tile_off = (int)tiles.size();
        L.ntx = (L.w + BLUR_TW - 1) / BLUR_TW; L.nty = (L.h + 4 * BLUR_TH - 1) / (4 * BLUR_TH);
        for (int ty = 0; ty < L.nty; ty++) for (int tx = 0; tx < L.ntx; tx++) tiles.push_back(make_int4(l, tx, ty, 0));
    }
    P.blur_bytes = (off + 255) & ~(size_t)255;                                     // all levels
    P.pyr_bytes = ((size_t)P.lev[0].pitch * P.lev[0].h + 255) & ~(size_t)255;         // level 0: the input image
    P.lvl_bytes = (off - (size_t)P.lev[0].pitch * P.lev[0].h + 255) & ~(size_t)255;  // levels >= 1
    // Everything between the input image and the key points / descriptors is scratch: it exists for a CHUNK of the batch, and
    // orb_run walks the batch chunk by chunk (5.5 of ORB's 5.9 MB per 640x480 frame are scratch).
    // Chunk size: every chunk boundary costs a tail (k_octree is one wave per frame and level), so the chunk is as large as a scratch
    // budget of 48 GB allows: the whole batch at 640x480 (8192 resident frames: 45 GB; two chunks of 4096 cost 5 ms of a 194 ms step),
    // 1024 frames at 1280x960, where the memory buys resident frames instead.
    {
        const size_t per = P.lvl_bytes + P.blur_bytes + (size_t)cells.size() * HVO_CELL_CAP * 4 + (size_t)cand_total * 12 + (size_t)kp_total * 4;
        P.chunk = batch;
        while (P.chunk > 1024 && (size_t)P.chunk * per > ((size_t)48 << 30)) P.chunk = (P.chunk + 1) / 2;
    }
    { const char *e = getenv("HVO_ORB_CHUNK"); if (e && atoi(e) > 0) P.chunk = std::min(batch, atoi(e)); }
    P.ncells = (int)cells.size(); P.cand_total = cand_total; P.node_total = node_total; P.kp_total = kp_total;
    {   // k_octree's node slots are dynamic LDS
        const size_t lds = (size_t)P.oct_slot_cap * OCT_SLOT_BYTES;
        if (lds > 150 * 1024) return HVO_ERR_UNSUPPORTED;
        if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_octree), lds)) return HVO_ERR_HIP;
    }
    P.ntiles = (int)tiles.size();
    P.kp_cap = kp_total;       // >= sum(nfeat)+8*nlevels: never truncates the octree output
    const size_t B = (size_t)batch;
    int rc;
    std::vector<LevelGeom> levv(P.lev, P.lev + nl);
    if ((rc = dev_upload(ctx, &P.d_lev, levv))) return rc;
    if ((rc = dev_upload(ctx, &P.d_cells, cells))) return rc;
    if ((rc = dev_upload(ctx, &P.d_rs_xofs, xofs))) return rc;
    if ((rc = dev_upload(ctx, &P.d_rs_xalpha, xalpha))) return rc;
    if ((rc = dev_upload(ctx, &P.d_rs_yofs, yofs))) return rc;
    if ((rc = dev_upload(ctx, &P.d_rs_ybeta, ybeta))) return rc;
    if ((rc = dev_upload(ctx, &P.d_tiles, tiles))) return rc;
    {   // the fused per-level pass (orb_level.hip); HVO_ORB_FUSED=0 keeps the three separate kernels
        std::vector<OrbTile> lt;
        const bool want = !(getenv("HVO_ORB_FUSED") && atoi(getenv("HVO_ORB_FUSED")) == 0);
        P.fused = want && orb_level_build(P, cells, xofs, yofs, lt);
        if (getenv("HVO_ORB_TPW")) P.lt_tpw = std::max(1, atoi(getenv("HVO_ORB_TPW")));
        if (getenv("HVO_ORB_NW")) { const int nw = atoi(getenv("HVO_ORB_NW")); P.lt_nw = nw == 1 || nw == 2 ? nw : 4; }
        if (P.fused && (rc = dev_upload(ctx, &P.d_ltiles, lt))) return rc;
    }
    if ((rc = orb_describe_build(ctx))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_pyr_base, B * P.pyr_bytes + 512))) return rc;
    P.d_pyr = P.d_pyr_base + 256;
    const size_t CB = (size_t)P.chunk;
    if ((rc = dev_alloc(ctx, &P.d_lvl_base, CB * P.lvl_bytes + 512))) return rc;
    P.d_lvl = P.d_lvl_base + 256;
    if ((rc = dev_alloc(ctx, &P.d_blur, CB * P.blur_bytes + 256))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_cell_kp, CB * P.ncells * HVO_CELL_CAP))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_cell_cnt, CB * P.ncells))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_cand, CB * cand_total))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_keys, CB * cand_total))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_keys_tmp, CB * cand_total))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_lvl_kp, CB * kp_total))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_lvl_cnt, CB * nl))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_kp, B * P.kp_cap))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_desc, B * P.kp_cap * 32))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_nkp, B))) return rc;
    if ((rc = dev_alloc(ctx, &P.d_flags, B))) return rc;
    // stream-ordered fills: a null-stream hipMemset is not ordered against the non-blocking ctx stream
    HVO_HIP(hipMemsetAsync(P.d_pyr_base, 0, B * P.pyr_bytes + 512, ctx->stream));
    HVO_HIP(hipMemsetAsync(P.d_lvl_base, 0, CB * P.lvl_bytes + 512, ctx->stream));
    HVO_HIP(hipMemsetAsync(P.d_blur, 0, CB * P.blur_bytes + 256, ctx->stream));
    HVO_HIP(hipMemsetAsync(P.d_flags, 0, B * sizeof(int), ctx->stream));
    HVO_HIP(hipMemsetAsync(P.d_nkp, 0, B * sizeof(int), ctx->stream));
    HVO_HIP(hipDeviceSynchronize());     // also drains the null-stream table uploads above
    return HVO_OK;
}

int orb_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h, bool sync)
{
    const hipStream_t cs = ctx->stage_gray_dst ? ctx->s_stage_up : hvo_copy_stream(ctx, ctx->stream);
    int rc = orb_ensure_plan(ctx, w, h, std::max(n, ctx->p.max_batch));
    if (rc) return rc;
    OrbPlan &P = ctx->orb;
    uint8_t *const dst = ctx->stage_gray_dst ? ctx->stage_gray_dst : P.d_pyr;      // the resident level-0 slab, or the staging slab of a double-buffered batch
    for (int f = 0; f < n; f++) if (!in[f].gray) return HVO_ERR_INVALID_ARG;
    // Frames that are dense (stride == width == device pitch) go up in RUNS: consecutive frames evenly spaced in host memory are ONE 2-D
    // copy whose "rows" are whole frames.  A copy call costs ~20 us, which at one call per frame is the whole upload (2048 frames of
    // 307 KB: 20 GB/s against the link's 57, profiles/r03_pcie_raw.txt); a batch that cycles through k distinct host frames is k-spaced
    // runs -- a handful of calls.
    const bool dense = P.lev[0].pitch == w && !ctx->kn_upload_single.off();
    for (int f = 0; f < n;) {
        int run = 1;
        if (dense && in[f].gray_stride == w && f + 1 < n && in[f + 1].gray_stride == w) {
            const ptrdiff_t step = in[f + 1].gray - in[f].gray;
            if (step >= (ptrdiff_t)w * h) {
                while (f + run < n && in[f + run].gray_stride == w && in[f + run].gray - in[f + run - 1].gray == step) run++;
                if (run > 1) HVO_HIP(hipMemcpy2DAsync(dst + (size_t)f * P.pyr_bytes, P.pyr_bytes, in[f].gray, (size_t)step, (size_t)w * h, run, hipMemcpyHostToDevice, cs));
            }
        }
        if (run == 1)
            HVO_HIP(hipMemcpy2DAsync(dst + (size_t)f * P.pyr_bytes, P.lev[0].pitch, in[f].gray, in[f].gray_stride, w, h, hipMemcpyHostToDevice, cs));
        f += run;
    }
    if (sync) HVO_HIP(hipStreamSynchronize(cs));
    return HVO_OK;
}

int orb_run(hvo_ctx *ctx, int n)
{
    OrbPlan &P = ctx->orb;
    if (n < 1 || n > P.batch) return HVO_ERR_INVALID_ARG;
    hipStream_t st = ctx->stream;
    const int nl = P.nlevels;
    HVO_HIP(hipMemsetAsync(P.d_flags, 0, n * sizeof(int), st));
    // OpenCV fixed-point Gaussian taps for ksize 7, sigma 2: getGaussianKernel(CV_32F) * 256, rounded
    static int k7[4] = { 0, 0, 0, 0 };
    if (!k7[3]) {
        float cf[7]; double sum = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(-0.5 / 4.0 * x * x); sum += cf[i]; }
        sum = 1. / sum;
        for (int i = 0; i < 4; i++) k7[i] = round_half_even_f((float)(cf[i] * sum) * 256.f);
    }
    int id;
    // the batch, chunk by chunk: frames [c0, c0 + m) use the scratch slabs as their frames 0 .. m-1
    P.last_chunks = (n + P.chunk - 1) / P.chunk;
    for (int c0 = 0; c0 < n; c0 += P.chunk) {
    const int m = std::min(P.chunk, n - c0);
    const bool last = c0 + m >= n;
    const uint8_t *pyr0 = P.d_pyr + (size_t)c0 * P.pyr_bytes;          // level 0 of the chunk's first frame
    int *flags = P.d_flags + c0;
    if (P.fused) {
        // one launch per level: level l is read once into LDS tiles and gives its FAST corners, its blurred image and level l+1
        id = hvo_prof_begin(ctx, "orb_levels", st);
        int rl = orb_level_run(ctx, c0, m, st, k7[0], k7[1], k7[2], k7[3]);
        if (!rl && (ctx->readings & HVO_READING_BLUR_FLOAT)) rl = orb_blur_float_run(ctx, c0, m, st);
        hvo_prof_end(ctx, id);
        if (rl) return rl;
        if (last && ctx->ev_fast && !ctx->serialize) { HVO_HIP(hipEventRecord(ctx->ev_fast, st)); ctx->fast_recorded = true; }
    } else {
    id = hvo_prof_begin(ctx, "orb_pyramid", st);
    for (int l = 1; l < nl; l++) {
        const LevelGeom &S = P.lev[l - 1], &D = P.lev[l];
        dim3 blk(64, 4), grd((D.w + 255) / 256, (D.h + 4 * RESIZE_ROWS - 1) / (4 * RESIZE_ROWS), m);
        const uint8_t *sb = l == 1 ? pyr0 : P.d_lvl + S.lvl_off; const size_t ss = l == 1 ? P.pyr_bytes : P.lvl_bytes;
        if (P.resize_dw[l])
            hipLaunchKernelGGL(k_resize_dw, grd, blk, 0, st, sb, ss, P.d_lvl + D.lvl_off, P.lvl_bytes, S, D, P.d_rs_xofs + D.rs_off, P.d_rs_xalpha + D.rs_off,
                               P.d_rs_yofs + D.ry_off, P.d_rs_ybeta + D.ry_off);
        else
            hipLaunchKernelGGL(k_resize, grd, blk, 0, st, sb, ss, P.d_lvl + D.lvl_off, P.lvl_bytes, S, D, P.d_rs_xofs + D.rs_off, P.d_rs_xalpha + D.rs_off,
                               P.d_rs_yofs + D.ry_off, P.d_rs_ybeta + D.ry_off);
    }
    hvo_prof_end(ctx, id);
    // the blur (input: the pyramid; output: what k_brief samples) is no link of the chain FAST -> octree -> orient -> BRIEF: it goes
    // either before k_fast_cells or right behind it (blur_late), wherever the other streams' long kernels absorb it best
    const bool blur_late = ctx->orb_blur_late;
    auto run_blur = [&]() -> int {
        id = hvo_prof_begin(ctx, "orb_blur", st);
        int rb_ = HVO_OK;
        if (ctx->readings & HVO_READING_BLUR_FLOAT) rb_ = orb_blur_float_run(ctx, c0, m, st);
        else hipLaunchKernelGGL(k_blur7, dim3(P.ntiles, m), dim3(256), 0, st, pyr0, P.pyr_bytes, P.d_lvl, P.lvl_bytes, P.d_blur, P.blur_bytes, P.d_lev, P.d_tiles, k7[0], k7[1], k7[2], k7[3]);
        hvo_prof_end(ctx, id);
        return rb_;
    };
    if (!blur_late) { const int rb = run_blur(); if (rb) return rb; }
    id = hvo_prof_begin(ctx, "orb_fast_cells", st);
    if (P.max_cell <= 45)
        hipLaunchKernelGGL(k_fast_cells<48>, dim3((P.ncells + 3) / 4, m), dim3(256), 0, st, pyr0, P.pyr_bytes, P.d_lvl, P.lvl_bytes, P.d_lev, P.d_cells, P.ncells,
                           P.d_cell_kp, P.d_cell_cnt, ctx->p.orb_ini_th_fast, ctx->p.orb_min_th_fast, flags);
    else
        hipLaunchKernelGGL(k_fast_cells<HVO_CELL_TILE>, dim3((P.ncells + 3) / 4, m), dim3(256), 0, st, pyr0, P.pyr_bytes, P.d_lvl, P.lvl_bytes, P.d_lev, P.d_cells, P.ncells,
                           P.d_cell_kp, P.d_cell_cnt, ctx->p.orb_ini_th_fast, ctx->p.orb_min_th_fast, flags);
    hvo_prof_end(ctx, id);
    if (last && ctx->ev_fast && !ctx->serialize) { HVO_HIP(hipEventRecord(ctx->ev_fast, st)); ctx->fast_recorded = true; }
    if (blur_late) { const int rb = run_blur(); if (rb) return rb; }
    }
    id = hvo_prof_begin(ctx, "orb_octree", st);
    OctArgs oa;
    oa.lev = P.d_lev; oa.cell_kp = P.d_cell_kp; oa.cell_cnt = P.d_cell_cnt; oa.ncells = P.ncells;
    oa.cand = P.d_cand; oa.keys = P.d_keys; oa.keys_tmp = P.d_keys_tmp; oa.lvl_kp = P.d_lvl_kp; oa.lvl_cnt = P.d_lvl_cnt; oa.flags = flags;
    oa.cand_total = P.cand_total; oa.kp_total = P.kp_total; oa.nlevels = nl; oa.slot_cap = P.oct_slot_cap;
    hipLaunchKernelGGL(k_octree, dim3(nl, m), dim3(64), (size_t)P.oct_slot_cap * OCT_SLOT_BYTES, st, oa);
    hvo_prof_end(ctx, id);
    { const int rd = orb_describe_run(ctx, c0, m, st); if (rd) return rd; }
    }   // chunks
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

int orb_download(hvo_ctx *ctx, int n, hvo_frame_out *out)
{
    const hipStream_t cs = hvo_copy_stream(ctx, ctx->stream);
    OrbPlan &P = ctx->orb;
    std::vector<int> nk(n), fl(n);
    HVO_HIP(hipMemcpyAsync(nk.data(), P.d_nkp, n * sizeof(int), hipMemcpyDeviceToHost, cs));
    HVO_HIP(hipMemcpyAsync(fl.data(), P.d_flags, n * sizeof(int), hipMemcpyDeviceToHost, cs));
    HVO_HIP(hipStreamSynchronize(cs));
    std::vector<void *> dk(n, nullptr), dd(n, nullptr); std::vector<size_t> bk(n, 0), bd(n, 0);
    for (int f = 0; f < n; f++) {
        int m = nk[f];
        if (fl[f]) out[f].status = HVO_ERR_CAPACITY;
        if (out[f].kp) {
            if (m > out[f].kp_cap) { m = out[f].kp_cap; out[f].status = HVO_ERR_CAPACITY; }
            if (m > 0) {
                dk[f] = out[f].kp; bk[f] = (size_t)m * sizeof(hvo_keypoint);
                if (out[f].desc) { dd[f] = out[f].desc; bd[f] = (size_t)m * 32; }
            }
        }
        out[f].n_kp = m;
    }
    int rc = hvo_staged_d2h(ctx, cs, P.d_kp, (size_t)P.kp_cap * sizeof(hvo_keypoint), n, dk.data(), bk.data());
    if (rc) return rc;
    if ((rc = hvo_staged_d2h(ctx, cs, P.d_desc, (size_t)P.kp_cap * 32, n, dd.data(), bd.data()))) return rc;
    HVO_HIP(hipStreamSynchronize(cs));
    return HVO_OK;
}

// diagnostics (not part of include/hvo.h): which ORB path the current plan runs and how it walks a batch -- out4 = { fused per-level pass
// (orb_level.hip) 1 / the separate kernels 0, frames per chunk, frames the plan holds, chunks the last orb_run walked }
extern "C" int hvo_debug_orb_plan(hvo_ctx *ctx, int *out4)
{
    if (!ctx || !out4 || ctx->orb.w <= 0) return HVO_ERR_INVALID_ARG;
    out4[0] = ctx->orb.fused ? 1 : 0; out4[1] = ctx->orb.chunk; out4[2] = ctx->orb.batch; out4[3] = ctx->orb.last_chunks;
    return HVO_OK;
}

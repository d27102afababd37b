// orb_level.hip -- one fused pass per pyramid level (the "pyramid+BRIEF pass" of BASELINE.json's north star).
//
// k_orb_level reads a level ONCE, tile by tile, into LDS and produces from that tile
//   * the FAST corners of the reference's cell whose interior the tile holds  (ORBextractor.cc:787-827; per-cell cv::FAST + fallback)
//   * the 7x7-blurred level                                                    (ORBextractor.cc:1083-1084, GaussianBlur 7x7 sigma 2)
//   * its share of the next pyramid level                                      (ORBextractor.cc:1105-1130, cv::resize INTER_LINEAR u8)
// instead of three kernels (k_resize_dw, k_fast_cells, k_blur7 of orb.hip) that each fetch the level from HBM.  One launch per level
// (level l+1 is an output of launch l), wave = tile, no workgroup barrier: the four waves of a workgroup work on neighbouring tiles
// of one frame, and all workgroups of a frame are dealt to the same XCD (blocks b and b+8 share one) so that the halo lines two
// tiles share are fetched into one L2 once.
//
// Tiles (host table, orb_level_build): the level is cut by the reference's own FAST grid -- a cell's interior [iniX+3, maxX-3) x
// [iniY+3, maxY-3) is a tile's FAST interior -- plus margin tiles for the 19-pixel border the cells do not cover.  A tile OWNS a
// rectangle of the level (4-aligned in x, so blurred output is whole dwords) for blur and resize; it loads interior and owned rectangle
// plus a 3-pixel halo (FAST ring radius = blur radius = 3; bilinear needs +1), REFLECT_101 at the image border as the reference's apron.
//
// Arithmetic is the oracle's, integer throughout:
//   FAST     score S = max over the 16 arcs of 9 of max(min d, -max d) - 1 (threshold-free; serves iniThFAST and the minThFAST fallback);
//            the pair pre-test runs on dwords of 4 pixels, survivors are queued in raster order and finished on full waves
//   blur     taps {18,34,49,55,49,34,18}/256 twice; rows by v_dot4_u32_u8 on shifted dwords, row sums (< 65536) stored in LDS as
//            VERTICAL u16 pairs so that the column pass is four v_dot2_u32_u16 per pixel; rounding of OpenCV 3.2's SymmColumnVec_32s8u
//   resize   11-bit coefficients, (b0*(t0>>4))>>16 + (b1*(t1>>4))>>16 + 2 >> 2
#include "hvo_internal.hpp"
#include <string.h>
#include <algorithm>
#include <vector>

#define LT_TP 64                    // LDS tile pitch in bytes (16 dwords = four 16-byte chunks, loaded as dwordx4)
#define LT_ND (LT_TP / 4)
#define LT_TR 46                    // tile rows (owned <= 40 + 6)
#define LT_MAXO 40                  // owned rectangle <= 40 x 40
#define LT_CAND 368
#define LT_PEND 328                 // 63 left over + 256 new + the trash slot
#define LT_TABN 48

typedef unsigned short lt_us2 __attribute__((ext_vector_type(2)));

// 8176 bytes per wave: five workgroups of four waves per CU (160 KB of LDS), five waves per SIMD
struct __attribute__((aligned(16))) LtWave {
    uint32_t T[LT_TR * LT_ND];                             // the level tile, rows of LT_ND dwords, 4-byte phase of global memory
    union {
        struct { uint8_t S[LT_TR * LT_TP]; unsigned short cand[LT_CAND]; unsigned short pend[LT_PEND]; unsigned long long mk_min[8], mk_ini[8]; } f;   // FAST: scores, lists
        uint32_t V[((LT_TR + 1) / 2) * LT_MAXO];           // then the blur's row-sum pairs
    } u;
    int tab[4][LT_TABN];                                  // next level's xofs / xalpha (from column dxa on) and yofs / ybeta (from row dya on) of this tile
};
static_assert(sizeof(LtWave) <= 8192, "five workgroups per CU");

struct LevelArgs {
    const uint8_t *rd; size_t rd_stride;               // the level read (frame 0 of the chunk): level 0 lives in the per-frame input slab, the others in the chunk's scratch
    uint8_t *wr; size_t wr_stride;                     // the next level
    uint8_t *blur; size_t blur_stride;                 // this level inside the all-levels blurred slab
    LevelGeom L, D; int has_next;
    const OrbTile *tiles; int ntiles, tpw, groups, nframes;
    const int *xofs, *xalpha, *yofs, *ybeta;
    uint32_t *cell_kp; int *cell_cnt; int ncells, iniTh, minTh; int *flags;
    int k0, k1, k2, k3;
    int res_dw;                         // the next level's taps fit the three-dword window (OrbPlan::resize_dw)
    int skip;                           // timing experiments only (HVO_LT_SKIP): 1 no FAST, 2 no blur, 4 no resize
};

static __device__ __forceinline__ int lt_reflect(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p < 0 ? 0 : p;
}
static __device__ __forceinline__ int lt_mbcnt(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

static __device__ __forceinline__ int lt_min3(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
static __device__ __forceinline__ int lt_max3(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// cornerScore<16> (OpenCV 3.2) on the LDS tile; v = centre.  With d = v - ring and e = ring - v the score is
// max over the 16 nine-arcs of max(min d, min e) - 1; a nine-arc minimum is a min3 of three min3 (explicit v_min3_i32: the
// compiler's reassociation of the min chains yields twice the instructions).
static __device__ __forceinline__ int lt_fast_score(const uint8_t *p, int v)
{
    constexpr int tp = LT_TP;
    int r[16];
    r[0] = p[3 * tp];    r[1] = p[3 * tp + 1];   r[2] = p[2 * tp + 2];   r[3] = p[tp + 3];
    r[4] = p[3];         r[5] = p[-tp + 3];      r[6] = p[-2 * tp + 2];  r[7] = p[-3 * tp + 1];
    r[8] = p[-3 * tp];   r[9] = p[-3 * tp - 1];  r[10] = p[-2 * tp - 2]; r[11] = p[-tp - 3];
    r[12] = p[-3];       r[13] = p[tp - 3];      r[14] = p[2 * tp - 2];  r[15] = p[3 * tp - 1];
    int d[16], e[16], a3[16], b3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { d[k] = v - r[k]; e[k] = r[k] - v; }
#pragma unroll
    for (int k = 0; k < 16; k++) { a3[k] = lt_min3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]); b3[k] = lt_min3(e[k], e[(k + 1) & 15], e[(k + 2) & 15]); }
    int best = -1000;
#pragma unroll
    for (int k = 0; k < 16; k++)
        best = lt_max3(best, lt_min3(a3[k], a3[(k + 3) & 15], a3[(k + 6) & 15]), lt_min3(b3[k], b3[(k + 3) & 15], b3[(k + 6) & 15]));
    return best - 1;
}

static __device__ __forceinline__ void lt_row_sums(unsigned W0, unsigned W1, unsigned W2, unsigned klo, unsigned khi, unsigned hs[4])
{
    // pixel j of dword W1 reads window bytes j+1 .. j+7 of {W0, W1, W2}
    const unsigned A0 = __builtin_amdgcn_alignbyte(W1, W0, 1), B0 = __builtin_amdgcn_alignbyte(W2, W1, 1);
    const unsigned A1 = __builtin_amdgcn_alignbyte(W1, W0, 2), B1 = __builtin_amdgcn_alignbyte(W2, W1, 2);
    const unsigned A2 = __builtin_amdgcn_alignbyte(W1, W0, 3), B2 = __builtin_amdgcn_alignbyte(W2, W1, 3);
    hs[0] = __builtin_amdgcn_udot4(A0, klo, __builtin_amdgcn_udot4(B0, khi, 0u, false), false);
    hs[1] = __builtin_amdgcn_udot4(A1, klo, __builtin_amdgcn_udot4(B1, khi, 0u, false), false);
    hs[2] = __builtin_amdgcn_udot4(A2, klo, __builtin_amdgcn_udot4(B2, khi, 0u, false), false);
    hs[3] = __builtin_amdgcn_udot4(W1, klo, __builtin_amdgcn_udot4(W2, khi, 0u, false), false);
}
static __device__ __forceinline__ unsigned lt_dot2(unsigned a, unsigned w, unsigned c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(lt_us2, a), __builtin_bit_cast(lt_us2, w), c, false);
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 5) void k_orb_level(const LevelArgs A)
{
    __shared__ LtWave lds_[NW];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // blocks b and b + 8 share an XCD: all groups of a frame go to one XCD
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int fr8 = q / A.groups, g = q - fr8 * A.groups;
    const int frame = fr8 * 8 + xcd;
    if (frame >= A.nframes) return;
    LtWave &W = lds_[wv];
    const LevelGeom &L = A.L;
    const uint8_t *img = A.rd + (size_t)frame * A.rd_stride;
    uint8_t *bdst = A.blur + (size_t)frame * A.blur_stride;
    const uint8_t *T8 = reinterpret_cast<const uint8_t *>(W.T);
    const unsigned long long ltm = (1ull << lane) - 1ull;

    for (int ti = 0; ti < A.tpw; ti++) {
        const int tidx = (g * A.tpw + ti) * NW + wv;
        if (tidx >= A.ntiles) break;
        const OrbTile t = A.tiles[tidx];
        const int lx0 = t.bx0 - 4, ly0 = t.by0 - 3;
        const int rows = t.bh + 6;
        const int xr = max(t.fx0 + t.fw, t.bx0 + t.bw) + 3;             // one past the last level column needed
        __builtin_amdgcn_wave_barrier();
        // ---- phase 0: the tile, 16-byte loads in the 4-byte phase of global memory, all in flight before the first is used; rows
        // reflected at the top / bottom border.  Columns outside the row are loaded as they come (the pyramid slab has a guard in
        // front and slack behind): the border fix-up below replaces the three that are used.  The next level's coefficient
        // tables for this tile's destination pixels ride along into LDS.
        {
            typedef uint32_t lt_u4 __attribute__((ext_vector_type(4), aligned(4)));
            uint4 v[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int i = lane + 64 * k, r = i >> 2, c = i & 3;
                v[k] = make_uint4(0, 0, 0, 0);
                if (r < rows) {
                    const int y = lt_reflect(ly0 + r, L.h);
                    const lt_u4 q4 = *reinterpret_cast<const lt_u4 *>(img + (size_t)y * L.pitch + lx0 + 16 * c);
                    v[k] = make_uint4(q4.x, q4.y, q4.z, q4.w);
                }
            }
            int tb0 = 0, tb1 = 0, tb2 = 0, tb3 = 0;
            const bool want_tab = A.has_next && lane < LT_TABN;
            if (want_tab) {
                const int dx = min(t.dxa + lane, A.D.w - 1), dy = min(t.dya + lane, A.D.h - 1);
                tb0 = A.xofs[dx]; tb1 = A.xalpha[dx]; tb2 = A.yofs[dy]; tb3 = A.ybeta[dy];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int i = lane + 64 * k;
                if ((i >> 2) < rows) reinterpret_cast<uint4 *>(W.T)[i] = v[k];
            }
            if (want_tab) { W.tab[0][lane] = tb0; W.tab[1][lane] = tb1; W.tab[2][lane] = tb2; W.tab[3][lane] = tb3; }
        }
        if (lx0 < 0) {                                                 // left border: x = -3..-1 <- x = 3, 2, 1 (tile bytes 1..3 <- 7, 6, 5)
            if (lane < rows) { const unsigned d1 = W.T[lane * LT_ND + 1]; W.T[lane * LT_ND] = __builtin_amdgcn_perm(d1, d1, 0x01020300u); }
        }
        if (xr > L.w) {                                                // right border: x = w + k <- w - 2 - k
            uint8_t *Tb = reinterpret_cast<uint8_t *>(W.T);
            if (lane < rows) {
                const int c = L.w - lx0;
#pragma unroll
                for (int k = 0; k < 3; k++) if (L.w + k < xr) Tb[lane * LT_TP + c + k] = Tb[lane * LT_TP + c - 2 - k];
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- phase 1: FAST-9-16 of the reference's cell (interior columns [fx0, fx0+fw), rows = the owned rows) ----
        if (t.fw > 0 && !(A.skip & 1)) {
            uint8_t *S = W.u.f.S;
            {
                uint4 *Sz = reinterpret_cast<uint4 *>(S);
                const int nz = (rows * LT_TP + 15) >> 4;
                for (int i = lane; i < nz; i += 64) Sz[i] = make_uint4(0, 0, 0, 0);
            }
            const int cx0 = t.fx0 - lx0, cx1 = cx0 + t.fw, ih = t.bh, iw = t.fw, P = iw * ih;
            const int dws = cx0 >> 2, nd = ((cx1 - 1) >> 2) - dws + 1, items = ih * nd;
            const int rcp = (65536 + nd - 1) / nd;
            const int minTh = A.minTh;
            int ncand = 0, npend = 0;
            // all eight opposite pairs for pend[0, cnt).  With cls = 1 (darker than v - t) | 2 (brighter than v + t) the test cv::FAST
            // itself uses is AND over the pairs of (cls[k] | cls[k+8]) != 0; per chain: every pair has a darker pixel  <=>  the LARGEST
            // of the pair minima is below v - t, every pair has a brighter one  <=>  the SMALLEST of the pair maxima is above v + t.
            auto rest = [&](const unsigned short *pp, int cnt) {
                int pass = 0, xy = 0;
                if (lane < cnt) {
                    xy = pp[lane];
                    const uint8_t *qq = T8 + (xy >> 8) * LT_TP + (xy & 0xFF);
                    const int v = qq[0];
#define LT_PMN(o) min((int)qq[o], (int)qq[-(o)])
#define LT_PMX(o) max((int)qq[o], (int)qq[-(o)])
                    const int n0 = LT_PMN(3 * LT_TP), n1 = LT_PMN(3), n2 = LT_PMN(2 * LT_TP + 2), n3 = LT_PMN(-2 * LT_TP + 2);
                    const int n4 = LT_PMN(3 * LT_TP + 1), n5 = LT_PMN(LT_TP + 3), n6 = LT_PMN(-LT_TP + 3), n7 = LT_PMN(-3 * LT_TP + 1);
                    const int x0 = LT_PMX(3 * LT_TP), x1 = LT_PMX(3), x2 = LT_PMX(2 * LT_TP + 2), x3 = LT_PMX(-2 * LT_TP + 2);
                    const int x4 = LT_PMX(3 * LT_TP + 1), x5 = LT_PMX(LT_TP + 3), x6 = LT_PMX(-LT_TP + 3), x7 = LT_PMX(-3 * LT_TP + 1);
#undef LT_PMN
#undef LT_PMX
                    const int M1 = lt_max3(lt_max3(n0, n1, n2), lt_max3(n3, n4, n5), max(n6, n7));
                    const int M2 = lt_min3(lt_min3(x0, x1, x2), lt_min3(x3, x4, x5), min(x6, x7));
                    pass = (int)(M1 < v - minTh) | (int)(M2 > v + minTh);
                }
                const unsigned long long m = __ballot(pass != 0);
                const int p = ncand + __popcll(m & ltm);
                if (pass && p < LT_CAND) W.u.f.cand[p] = (unsigned short)xy;
                ncand += __popcll(m);
            };
            // stage 1: the vertical pair 0 | 8 on dwords of four pixels (no short-circuit: every lane runs the same few instructions);
            // survivors are queued in raster order and finished by rest() on full waves
            for (int base = 0; base < items; base += 64) {
                const int it = base + lane;
                const int ok = it < items;
                const int itc = ok ? it : 0;
                const int r = (itc * rcp) >> 16, dwi = dws + itc - r * nd;
                const unsigned t4 = W.T[r * LT_ND + dwi], v4 = W.T[(r + 3) * LT_ND + dwi], b4 = W.T[(r + 6) * LT_ND + dwi];
                const int xb = 4 * dwi;
                const int jl = max(cx0 - xb, 0), jh = min(cx1 - xb, 4);                       // valid bytes [jl, jh)
                const unsigned inm = ok ? (((1u << jh) - 1u) & ~((1u << jl) - 1u)) : 0u;
                int pj[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = (v4 >> (8 * j)) & 0xFF, tt = (t4 >> (8 * j)) & 0xFF, bb = (b4 >> (8 * j)) & 0xFF;
                    pj[j] = (int)((inm >> j) & 1u) & ((int)(min(tt, bb) < v - minTh) | (int)(max(tt, bb) > v + minTh));
                }
                const unsigned long long m0 = __ballot(pj[0] != 0), m1 = __ballot(pj[1] != 0), m2 = __ballot(pj[2] != 0), m3 = __ballot(pj[3] != 0);
                if ((m0 | m1 | m2 | m3) != 0ull) {
                    const int p0 = npend + lt_mbcnt(m0) + lt_mbcnt(m1) + lt_mbcnt(m2) + lt_mbcnt(m3);
                    const int p1 = p0 + pj[0], p2 = p1 + pj[1], p3 = p2 + pj[2];
                    const int yx = ((r + 3) << 8) | xb;
                    // rejected pixels write to a slot nobody reads (no exec-mask games)
                    W.u.f.pend[pj[0] ? p0 : LT_PEND - 1] = (unsigned short)yx;
                    W.u.f.pend[pj[1] ? p1 : LT_PEND - 1] = (unsigned short)(yx + 1);
                    W.u.f.pend[pj[2] ? p2 : LT_PEND - 1] = (unsigned short)(yx + 2);
                    W.u.f.pend[pj[3] ? p3 : LT_PEND - 1] = (unsigned short)(yx + 3);
                    npend += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
                    int head = 0;
                    while (npend - head >= 64) { rest(W.u.f.pend + head, 64); head += 64; }
                    if (head) {
                        const int left = npend - head;
                        const int tv = W.u.f.pend[head + (lane < left ? lane : 0)];
                        if (lane < left) W.u.f.pend[lane] = (unsigned short)tv;
                        npend = left;
                    }
                }
            }
            if (npend > 0) rest(W.u.f.pend, npend);
            __builtin_amdgcn_wave_barrier();
            // exact scores of the candidates (pathological tile whose list overflowed: of every interior pixel)
            const uint8_t *T0 = T8 + 3 * LT_TP + cx0;                   // T0[y*TP + x] = interior pixel (x, y)
            uint8_t *S0 = S + 3 * LT_TP + cx0;
            if (ncand <= LT_CAND) {
                for (int i = lane; i < ncand; i += 64) {
                    const int xy = W.u.f.cand[i], o = (xy >> 8) * LT_TP + (xy & 0xFF);
                    const int s = lt_fast_score(T8 + o, T8[o]);
                    S[o] = (uint8_t)(s >= minTh ? s : 0);
                }
            } else {
                for (int p = lane; p < P; p += 64) {
                    const int y = p / iw, x = p - y * iw;
                    const int s = lt_fast_score(T0 + y * LT_TP + x, T0[y * LT_TP + x]);
                    S0[y * LT_TP + x] = (uint8_t)(s >= minTh ? s : 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // strict 3x3 NMS inside the interior (scores outside it are never written: 0, like the reference's per-view FAST call) and
            // emission in raster order; iniThFAST survivors if there are any, else the minThFAST ones (ORBextractor.cc:809-816)
            uint32_t *out = A.cell_kp + ((size_t)frame * A.ncells + t.cell) * HVO_CELL_CAP;
            const int ex = lx0 - L.minBX, ey = ly0 - L.minBY;           // emitted coordinates are relative to (minBorderX, minBorderY)
            int n_ini = 0, pos = 0;
            if (ncand <= LT_CAND) {
                int st = 0;
                for (int base = 0; base < ncand; base += 64, st++) {
                    const int i = base + lane;
                    bool okk = false; int v = 0;
                    if (i < ncand) {
                        const int xy = W.u.f.cand[i];
                        const uint8_t *sp = S + (xy >> 8) * LT_TP + (xy & 0xFF);
                        v = sp[0];
                        const int nb = lt_max3(lt_max3(sp[-1], sp[1], sp[-LT_TP - 1]), lt_max3(sp[-LT_TP], sp[-LT_TP + 1], sp[LT_TP - 1]), max((int)sp[LT_TP], (int)sp[LT_TP + 1]));
                        okk = (v != 0) & (v > nb);
                    }
                    const unsigned long long mm = __ballot(okk), mi = __ballot(okk & (v >= A.iniTh));
                    n_ini += __popcll(mi);
                    if (lane == 0) { W.u.f.mk_min[st] = mm; W.u.f.mk_ini[st] = mi; }
                }
                __builtin_amdgcn_wave_barrier();
                const bool use_ini = n_ini > 0;
                st = 0;
                for (int base = 0; base < ncand; base += 64, st++) {
                    const unsigned long long m = use_ini ? W.u.f.mk_ini[st] : W.u.f.mk_min[st];
                    if (m) {
                        if ((m >> lane) & 1ull) {
                            const int xy = W.u.f.cand[base + lane], x = xy & 0xFF, y = xy >> 8;
                            const int p = pos + __popcll(m & ltm);
                            if (p < HVO_CELL_CAP) out[p] = (uint32_t)(x + ex) | ((uint32_t)(y + ey) << 12) | ((uint32_t)S[y * LT_TP + x] << 24);
                        }
                        pos += __popcll(m);
                    }
                }
            } else {
                // full-raster walk, two passes (the count of iniThFAST survivors decides which set is emitted)
                for (int pass = 0; pass < 2; pass++) {
                    const bool use_ini = n_ini > 0;
                    for (int p0 = 0; p0 < P; p0 += 64) {
                        const int p = p0 + lane;
                        bool okk = false; int v = 0, x = 0, y = 0;
                        if (p < P) {
                            y = p / iw; x = p - y * iw;
                            const uint8_t *sp = S0 + y * LT_TP + x;
                            v = sp[0];
                            const int nb = lt_max3(lt_max3(sp[-1], sp[1], sp[-LT_TP - 1]), lt_max3(sp[-LT_TP], sp[-LT_TP + 1], sp[LT_TP - 1]), max((int)sp[LT_TP], (int)sp[LT_TP + 1]));
                            okk = (v != 0) & (v > nb);
                        }
                        if (pass == 0) n_ini += __popcll(__ballot(okk && v >= A.iniTh));
                        else {
                            const bool em = okk && (!use_ini || v >= A.iniTh);
                            const unsigned long long m = __ballot(em);
                            if (em) {
                                const int pp = pos + __popcll(m & ltm);
                                if (pp < HVO_CELL_CAP) out[pp] = (uint32_t)(x + cx0 + ex) | ((uint32_t)(y + 3 + ey) << 12) | ((uint32_t)v << 24);
                            }
                            pos += __popcll(m);
                        }
                    }
                }
            }
            if (lane == 0) {
                if (pos > HVO_CELL_CAP) { atomicOr(&A.flags[frame], 1); pos = HVO_CELL_CAP; }
                A.cell_cnt[(size_t)frame * A.ncells + t.cell] = pos;
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- phase 2: 7x7 Gaussian of the owned rectangle ----
        const int nbd = t.bw >> 2;
        if (nbd > 0 && !(A.skip & 2)) {
            const unsigned klo = (unsigned)A.k0 | ((unsigned)A.k1 << 8) | ((unsigned)A.k2 << 16) | ((unsigned)A.k3 << 24);
            const unsigned khi = (unsigned)A.k2 | ((unsigned)A.k1 << 8) | ((unsigned)A.k0 << 16);
            const int np = (rows + 1) >> 1;
            const int rcpb = (65536 + nbd - 1) / nbd;
            for (int it = lane; it < np * nbd; it += 64) {
                const int p = (it * rcpb) >> 16, o = it - p * nbd;
                const int r0 = 2 * p, r1 = min(2 * p + 1, rows - 1);
                const uint32_t *ra = W.T + r0 * LT_ND + o, *rb = W.T + r1 * LT_ND + o;      // dwords o, o+1, o+2: owned dword is o+1
                unsigned ha[4], hb[4];
                lt_row_sums(ra[0], ra[1], ra[2], klo, khi, ha);
                lt_row_sums(rb[0], rb[1], rb[2], klo, khi, hb);
                *reinterpret_cast<uint4 *>(W.u.V + p * LT_MAXO + 4 * o) =
                    make_uint4(ha[0] | (hb[0] << 16), ha[1] | (hb[1] << 16), ha[2] | (hb[2] << 16), ha[3] | (hb[3] << 16));
            }
            __builtin_amdgcn_wave_barrier();
            // column pass: output rows y (even) and y+1 from the four row pairs y/2 .. y/2+3
            const unsigned k01 = (unsigned)A.k0 | ((unsigned)A.k1 << 16), k23 = (unsigned)A.k2 | ((unsigned)A.k3 << 16);
            const unsigned k21 = (unsigned)A.k2 | ((unsigned)A.k1 << 16), k0_ = (unsigned)A.k0;
            const unsigned k_0 = (unsigned)A.k0 << 16, k12 = (unsigned)A.k1 | ((unsigned)A.k2 << 16);
            const unsigned k32 = (unsigned)A.k3 | ((unsigned)A.k2 << 16), k10 = (unsigned)A.k1 | ((unsigned)A.k0 << 16);
            const int nq = (t.bh + 1) >> 1;
            const int wv4 = L.w & ~3;
            for (int it = lane; it < nq * nbd; it += 64) {
                const int qq = (it * rcpb) >> 16, o = it - qq * nbd;
                const uint4 P0 = *reinterpret_cast<const uint4 *>(W.u.V + qq * LT_MAXO + 4 * o);
                const uint4 P1 = *reinterpret_cast<const uint4 *>(W.u.V + (qq + 1) * LT_MAXO + 4 * o);
                const uint4 P2 = *reinterpret_cast<const uint4 *>(W.u.V + (qq + 2) * LT_MAXO + 4 * o);
                const uint4 P3 = *reinterpret_cast<const uint4 *>(W.u.V + (qq + 3) * LT_MAXO + 4 * o);
                const unsigned p0[4] = { P0.x, P0.y, P0.z, P0.w }, p1[4] = { P1.x, P1.y, P1.z, P1.w };
                const unsigned p2[4] = { P2.x, P2.y, P2.z, P2.w }, p3[4] = { P3.x, P3.y, P3.z, P3.w };
                const int x0 = t.bx0 + 4 * o;
                // columns x < (w & ~3): round half to even of s / 65536 (the SSE2 float path, exact here); the last w % 4 columns: (s + 32768) >> 16
                const unsigned halfup = x0 >= wv4 ? 1u : 0u;
                unsigned ta[4], tb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned s0 = lt_dot2(p0[j], k01, lt_dot2(p1[j], k23, lt_dot2(p2[j], k21, lt_dot2(p3[j], k0_, 0u))));
                    const unsigned s1 = lt_dot2(p0[j], k_0, lt_dot2(p1[j], k12, lt_dot2(p2[j], k32, lt_dot2(p3[j], k10, 0u))));
                    ta[j] = min(s0 + 32767u + (((s0 >> 16) & 1u) | halfup), 0x00FFFFFFu);      // byte 2 = the rounded, saturated quotient
                    tb[j] = min(s1 + 32767u + (((s1 >> 16) & 1u) | halfup), 0x00FFFFFFu);
                }
                const unsigned oa = __builtin_amdgcn_perm(__builtin_amdgcn_perm(ta[3], ta[2], 0x0c0c0602u), __builtin_amdgcn_perm(ta[1], ta[0], 0x0c0c0602u), 0x05040100u);
                const unsigned ob = __builtin_amdgcn_perm(__builtin_amdgcn_perm(tb[3], tb[2], 0x0c0c0602u), __builtin_amdgcn_perm(tb[1], tb[0], 0x0c0c0602u), 0x05040100u);
                const int y = t.by0 + 2 * qq;
                *reinterpret_cast<uint32_t *>(bdst + (size_t)y * L.pitch + x0) = oa;
                if (2 * qq + 1 < t.bh) *reinterpret_cast<uint32_t *>(bdst + (size_t)(y + 1) * L.pitch + x0) = ob;
            }
        }

        // ---- phase 3: this tile's share of the next level: the destination DWORDS whose first pixel's left tap the tile owns (rows: whose
        // upper tap it owns).  The other three pixels' taps reach at most 5 columns past the owned rectangle: inside the 64-byte tile rows.
        if (A.has_next && t.dxb > t.dxa && t.dyb > t.dya && !(A.skip & 4)) {
            const LevelGeom &D = A.D;
            uint8_t *dst = A.wr + (size_t)frame * A.wr_stride;
            const int dd0 = t.dxa >> 2, ndd = (t.dxb - t.dxa) >> 2;          // dxa, dxb are multiples of 4
            const int nr = t.dyb - t.dya;
            constexpr int RG = 4;
            const int ngr = (nr + RG - 1) / RG;
            const int rcpd = (65536 + ndd - 1) / ndd;
            for (int it = lane; it < ngr * ndd; it += 64) {
                const int gr = (it * rcpd) >> 16, c = it - gr * ndd;
                const int dxq = 4 * (dd0 + c);
                int sxo[4]; unsigned a01[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    sxo[i] = W.tab[0][4 * c + i] - lx0;                   // tile column of the left tap
                    a01[i] = (unsigned)W.tab[1][4 * c + i];               // (a0 | a1 << 16), both in [0, 2048]
                }
                // the 8 source bytes of a row (4 pixels x 2 taps) lie within 9 bytes of an aligned dword when the scale factor is <= 4/3
                // (host-checked per level, as for k_resize_dw): three dwords, picked by v_perm_b32 with a selector fixed per lane; a pixel's
                // two taps become one (L | R << 16) and its horizontal pass one v_dot2_u32_u16 with (a0 | a1 << 16)
                const int wbase = sxo[0] >> 2;
                unsigned sel = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) sel |= (unsigned)((sxo[i] - 4 * wbase) & 7) << (8 * i);
                auto hrow = [&](int sy, unsigned tt[4]) {
                    if (A.res_dw) {
                        const uint32_t *rp = W.T + (sy - ly0) * LT_ND + wbase;
                        const uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2];
                        const uint32_t Lb = __builtin_amdgcn_perm(w1, w0, sel);
                        const uint32_t Rb = __builtin_amdgcn_perm(__builtin_amdgcn_alignbyte(w2, w1, 1), __builtin_amdgcn_alignbyte(w1, w0, 1), sel);
                        tt[0] = lt_dot2(__builtin_amdgcn_perm(Rb, Lb, 0x0c040c00u), a01[0], 0u);
                        tt[1] = lt_dot2(__builtin_amdgcn_perm(Rb, Lb, 0x0c050c01u), a01[1], 0u);
                        tt[2] = lt_dot2(__builtin_amdgcn_perm(Rb, Lb, 0x0c060c02u), a01[2], 0u);
                        tt[3] = lt_dot2(__builtin_amdgcn_perm(Rb, Lb, 0x0c070c03u), a01[3], 0u);
                    } else {
                        const uint8_t *rp = T8 + (sy - ly0) * LT_TP;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int sx = min(max(sxo[i], 0), LT_TP - 2);
                            tt[i] = lt_dot2((unsigned)rp[sx] | ((unsigned)rp[sx + 1] << 16), a01[i], 0u);
                        }
                    }
                };
                // every lane runs RG rows (rows past the tile's share recompute its last row and are not stored): no divergent loop
#pragma unroll
                for (int rr = 0; rr < RG; rr++) {
                    const int dyu = t.dya + gr * RG + rr, dy = min(dyu, t.dyb - 1);
                    const int yo = W.tab[2][dy - t.dya], sy0 = yo & 0xFFFF, sy1 = yo >> 16;
                    const unsigned yb = (unsigned)W.tab[3][dy - t.dya];
                    const unsigned b0 = yb & 0xFFFFu, b1 = yb >> 16;          // in [0, 2048]
                    unsigned t0[4], t1[4];
                    hrow(sy0, t0);
                    hrow(sy1, t1);
                    uint32_t ov = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const unsigned v = ((__umul24(b0, t0[i] >> 4) >> 16) + (__umul24(b1, t1[i] >> 4) >> 16) + 2u) >> 2;
                        ov |= (v & 0xFFu) << (8 * i);
                    }
                    if (dyu < t.dyb) *reinterpret_cast<uint32_t *>(dst + (size_t)dy * D.pitch + dxq) = ov;
                }
            }
        }
    }
}

// =====================================================================================
// host: the tile table of every level
// =====================================================================================
struct LtIv { int a, b, idx; };

static void lt_split(std::vector<LtIv> &v, int a, int b, int align)
{
    if (b <= a) return;
    const int len = b - a, n = (len + LT_MAXO - 1) / LT_MAXO;
    int step = (len + n - 1) / n;
    step = (step + align - 1) / align * align;
    for (int s = a; s < b; s += step) v.push_back({ s, std::min(s + step, b), -1 });
}

// Builds P.h_tiles / per-level offsets; returns false when a level's geometry does not fit the LDS tile (caller keeps the unfused kernels).
bool orb_level_build(OrbPlan &P, const std::vector<CellDesc> &cells, const std::vector<int> &xofs, const std::vector<int> &yofs, std::vector<OrbTile> &tiles)
{
    tiles.clear();
    for (int l = 0; l < P.nlevels; l++) {
        LevelGeom &L = P.lev[l];
        P.lt_off[l] = (int)tiles.size();
        // kept columns / rows of the FAST grid, from the cell table (a full product grid: the keep rules are separable)
        std::vector<LtIv> cols, rws;       // interiors
        for (int c = L.cell_off; c < L.cell_off + L.ncells; c++) {
            const CellDesc &cd = cells[c];
            const int fx = cd.x0 + 3, fxe = cd.x0 + cd.vw - 3, fy = cd.y0 + 3, fye = cd.y0 + cd.vh - 3;
            bool hc = false, hr = false;
            for (auto &iv : cols) hc |= iv.a == fx;
            for (auto &iv : rws) hr |= iv.a == fy;
            if (!hc) cols.push_back({ fx, fxe, (int)cols.size() });
            if (!hr) rws.push_back({ fy, fye, (int)rws.size() });
        }
        if ((int)(cols.size() * rws.size()) != L.ncells) return false;
        for (size_t j = 1; j < cols.size(); j++) if (cols[j].a != cols[j - 1].b) return false;
        for (size_t i = 1; i < rws.size(); i++) if (rws[i].a != rws[i - 1].b) return false;
        const int W4 = (L.w + 3) & ~3;
        std::vector<LtIv> xs, ys;          // owned intervals; idx = kept column / row or -1
        if (cols.empty()) { lt_split(xs, 0, W4, 4); lt_split(ys, 0, L.h, 1); }
        else {
            lt_split(xs, 0, cols[0].a & ~3, 4);
            for (size_t j = 0; j < cols.size(); j++) {
                const int a = cols[j].a & ~3, b = (j + 1 < cols.size() ? cols[j + 1].a : cols[j].b) & ~3;
                xs.push_back({ a, b, (int)j });
            }
            lt_split(xs, cols.back().b & ~3, W4, 4);
            lt_split(ys, 0, rws[0].a, 1);
            for (size_t i = 0; i < rws.size(); i++) ys.push_back({ rws[i].a, rws[i].b, (int)i });
            lt_split(ys, rws.back().b, L.h, 1);
        }
        const bool has_next = l + 1 < P.nlevels;
        const int *xo = has_next ? xofs.data() + P.lev[l + 1].rs_off : nullptr;
        const int *yo = has_next ? yofs.data() + P.lev[l + 1].ry_off : nullptr;
        const int dw = has_next ? P.lev[l + 1].w : 0, dh = has_next ? P.lev[l + 1].h : 0;
        for (auto &iy : ys) for (auto &ix : xs) {
            OrbTile t; memset(&t, 0, sizeof(t));
            t.bx0 = (short)ix.a; t.bw = (short)(ix.b - ix.a); t.by0 = (short)iy.a; t.bh = (short)(iy.b - iy.a);
            t.cell = -1; t.fx0 = (short)ix.a; t.fw = 0;
            if (ix.idx >= 0 && iy.idx >= 0) {
                t.fx0 = (short)cols[ix.idx].a; t.fw = (short)(cols[ix.idx].b - cols[ix.idx].a);
                t.cell = (short)(L.cell_off + iy.idx * (int)cols.size() + ix.idx);
                const CellDesc &cd = cells[t.cell];
                if (cd.x0 + 3 != t.fx0 || cd.y0 + 3 != t.by0 || cd.vh - 6 != t.bh || cd.vw - 6 != t.fw) return false;
            }
            if (t.bh < 1 || (t.bw < 1 && t.fw < 1)) continue;
            const int xr = std::max(t.fx0 + t.fw, t.bx0 + t.bw) + 3;
            if (t.bw > LT_MAXO || t.bh > LT_MAXO || t.bh + 6 > LT_TR || xr - (t.bx0 - 4) > LT_TP || (t.bw & 3) || (t.bx0 & 3)) return false;
            if (t.fw > 0 && ((t.fw * t.bh + 3) / 4 > LT_CAND + 1000)) return false;
            int dxa = 0, dxb = 0, dya = 0, dyb = 0;
            if (has_next && t.bw > 0) {
                // destination dwords whose FIRST pixel's left tap lies in the owned columns; rows whose upper tap lies in the owned rows
                const int dw4 = (dw + 3) & ~3;
                while (dxa < dw4 && xo[std::min(dxa, dw - 1)] < t.bx0) dxa += 4;
                dxb = dxa; while (dxb < dw4 && xo[std::min(dxb, dw - 1)] < t.bx0 + t.bw) dxb += 4;
                while (dya < dh && (yo[dya] & 0xFFFF) < t.by0) dya++;
                dyb = dya; while (dyb < dh && (yo[dyb] & 0xFFFF) < t.by0 + t.bh) dyb++;
                // the taps of the dword's other pixels must stay inside the 64-byte tile row (and left of the fixed-up border columns + 1)
                for (int dx = dxa; dx < std::min(dxb, dw); dx++) if (xo[dx] + 1 - (t.bx0 - 4) > LT_TP - 1) return false;
            }
            if (dxb - dxa > LT_TABN || dyb - dya > LT_TABN) return false;
            t.dxa = (short)dxa; t.dxb = (short)dxb; t.dya = (short)dya; t.dyb = (short)dyb;
            tiles.push_back(t);
        }
        P.lt_cnt[l] = (int)tiles.size() - P.lt_off[l];
    }
    return true;
}

// the eight launches of the fused pass (levels in order: launch l writes level l+1)
int orb_level_run(hvo_ctx *ctx, int c0, int n, hipStream_t st, int k0, int k1, int k2, int k3)
{
    OrbPlan &P = ctx->orb;
    for (int l = 0; l < P.nlevels; l++) {
        LevelArgs A;
        A.L = P.lev[l]; A.has_next = l + 1 < P.nlevels; A.D = P.lev[A.has_next ? l + 1 : l];
        if (l == 0) { A.rd = P.d_pyr + (size_t)c0 * P.pyr_bytes; A.rd_stride = P.pyr_bytes; } else { A.rd = P.d_lvl + A.L.lvl_off; A.rd_stride = P.lvl_bytes; }
        A.wr = P.d_lvl + A.D.lvl_off; A.wr_stride = P.lvl_bytes;
        A.blur = P.d_blur + A.L.img_off; A.blur_stride = P.blur_bytes;
        A.tiles = P.d_ltiles + P.lt_off[l]; A.ntiles = P.lt_cnt[l];
        const int nw = P.lt_nw;                            // waves per workgroup: read once when the plan was built (HVO_ORB_NW)
        A.tpw = P.lt_tpw; A.groups = (A.ntiles + nw * A.tpw - 1) / (nw * A.tpw); A.nframes = n;
        A.xofs = P.d_rs_xofs + A.D.rs_off; A.xalpha = P.d_rs_xalpha + A.D.rs_off; A.yofs = P.d_rs_yofs + A.D.ry_off; A.ybeta = P.d_rs_ybeta + A.D.ry_off;
        A.cell_kp = P.d_cell_kp; A.cell_cnt = P.d_cell_cnt; A.ncells = P.ncells; A.iniTh = ctx->p.orb_ini_th_fast; A.minTh = ctx->p.orb_min_th_fast;
        A.res_dw = A.has_next && P.resize_dw[l + 1];
#ifdef HVO_TIMING_KNOBS
        A.skip = getenv("HVO_LT_SKIP") ? atoi(getenv("HVO_LT_SKIP")) : 0;      // phases skipped for timing (tools/orb_phase_sweep.py): results are NOT valid outputs
#else
        A.skip = 0;                                        // (the phase-skip mask exists in -DHVO_TIMING_KNOBS builds only: a stray variable must not corrupt ORB output)
#endif
        if (ctx->readings & HVO_READING_BLUR_FLOAT) A.skip |= 2;      // the float-kernel reading of GaussianBlur: the blurred levels come from readings.hip (orb_blur_float_run)
        A.flags = P.d_flags + c0; A.k0 = k0; A.k1 = k1; A.k2 = k2; A.k3 = k3;
        if (A.ntiles < 1) continue;
        const int n8 = (n + 7) / 8 * 8;
        if (nw == 1) hipLaunchKernelGGL(k_orb_level<1>, dim3((unsigned)(n8 * A.groups)), dim3(64), 0, st, A);
        else if (nw == 2) hipLaunchKernelGGL(k_orb_level<2>, dim3((unsigned)(n8 * A.groups)), dim3(128), 0, st, A);
        else hipLaunchKernelGGL(k_orb_level<4>, dim3((unsigned)(n8 * A.groups)), dim3(256), 0, st, A);
    }
    return HVO_OK;
}

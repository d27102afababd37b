// peac.hip -- batched PEAC / AHC plane extraction for gfx950 (MI355X).
//
// Replaces PlaneDetection::readDepthImage + runPlaneDetection (reference src/PlaneExtractor.cpp:26-66)
// i.e. ahc::PlaneFitter::run (reference include/peac/AHCPlaneFitter.hpp:211-260) for a batch of frames.
//
//   k_peac_blocks    readDepthImage fused with the PlaneSeg block constructor + Stats::compute
//                    (PlaneExtractor.cpp:42-56, AHCPlaneSeg.hpp:210-285, 125-156): the 7.4 MB fp64
//                    cloud is never materialised, each 10x10 block is unprojected on the fly
//   k_peac_cluster   initGraph edges (AHCPlaneFitter.hpp:894-954) + ahCluster (983-1189); 16 lanes per
//                    frame (4 frames per wave in lockstep), 16-ary min-MSE heap in global memory touched
//                    once per iteration, candidate merges evaluated one per lane, unordered neighbour sets
//   k_peac_blkmap    findBlockMembership (485-587): block erosion, packed per-pixel flood state
//   k_peac_flood     seed queue + floodFill (428-476): a thread owns a queue entry and fetches its four neighbour
//                    states; the live third of the events is compacted and resolved per pixel group through an
//                    LDS hash (closed form for one-plane groups, ranked / serial replay for the rest)
//   k_peac_final     last merge round + plidmap (299-340), one wave per frame
//   k_peac_relabel   membership relabel (353-365), negative "trail" counters reported as -1
//
// fp64 throughout, identical operation order to oracle/peac.c (no FMA contraction: -ffp-contract=off).
// Tie rules where the reference is address dependent (std::set<PlaneSeg*>, priority_queue ties):
// node creation order, see oracle/peac.c header.
#include <type_traits>
#include "hvo_internal.hpp"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define WIN 10
#define MIN_SUPPORT 3000
#define MAX_PLANES 64
#define SEG_D 16          // doubles per seg: st[9], center[3], normal[3], mse
#define SEG_I 8           // ints per seg: N, rid, nouse, nb_off, nb_cnt, nb_cap, valid, pad
#define LCAP 1024         // neighbour-list length staged in LDS

struct PeacPlan {
    int w = 0, h = 0, pitch = 0, Nw = 0, Nh = 0, nblk = 0, segcap = 0, poolcap = 0, qcap = 0, batch = 0;
    uint16_t *d_depth = nullptr;
    double *d_segD = nullptr; int *d_segI = nullptr; int *d_pool = nullptr, *d_pool2 = nullptr;
    int *d_parent = nullptr, *d_dsize = nullptr, *d_eflag = nullptr;
    int *d_meta = nullptr;          // per frame 16 ints: [0]=nseg [1]=pooltop [2]=nextracted [3]=flags [4]=nfinal [5]=nq
    int *d_extracted = nullptr;     // per frame MAX_PLANES seg ids (coarse planes), then MAX_PLANES final
    int *d_blkmap = nullptr; int8_t *d_labels = nullptr; uint32_t *d_state = nullptr;     // labels: int8 on the device and on the wire (<= 64 planes), int32 at the ABI
    int *d_queue = nullptr; int *d_plidmap = nullptr; int *d_isvalid = nullptr;
    hvo_plane *d_planes = nullptr;
    unsigned long long *d_adj = nullptr;
    void *d_hot = nullptr;                                                 // HotNode records of the grouped AHC kernel
    double *d_hkey = nullptr, *d_m1k = nullptr; int *d_hid = nullptr;      // TQueue: keys, bucket minima, their ids
    int *d_lq = nullptr;                                                   // k_peac_cluster_slots: (label << 16 | slot) of every slot
    double c15 = 0, c60 = 0, c30 = 0;   // cos thresholds evaluated on the host (glibc), like the oracle
    double ang_factor = 0, ang_near = 0;
    // tuning variables, read when the plan is built (peac_build_plan)
    struct { Knob edges, gl, perm, lend, slots, heads_maxn, heads, heads_big, poolcap, ldsq, flood_t, flood_epl, flood_perm; } kn;
};

static PeacPlan *plan_of(hvo_ctx *ctx) { return (PeacPlan *)ctx->peac; }

// ------------------------------------------------------------------------------------------------
// smallest eigenpair of the 3x3 covariance -- mirrors oracle/peac.c orc_eig33_smallest operation by operation:
// Laguerre's iteration on det(l I - K) from l = 0 (monotone from below, 2-3 steps for plane-like patches, each one
// fp64 sqrt + one division), then the column of adj(K - l I) with the largest diagonal minor, normalised.
// ~3.5 sqrt + 3.5 div + 120 flops instead of the ~20 + 20 + 600 of the cyclic Jacobi solve it replaces, with the
// backward-stable accuracy (2e-16 trace) that the reference's Eigen solver has.
// ------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void eig33_smallest_dev(double a, double b, double c, double d, double e, double f, double &l0, double v[3])
{
    const double tol = 2.220446049250313e-16 * (a + b + c);
    double l = 0.0;
    bool go = true;
#pragma unroll 1
    for (int it = 0; it < 8; it++) {
        if (!__any(go)) break;                                 // the wave leaves together; finished lanes keep their l
        const double A = a - l, B = b - l, C = c - l;
        const double m1 = B * C - e * e, m2 = A * C - f * f, m3 = A * B - d * d;
        const double det = A * m1 - d * (d * C - e * f) + f * (d * e - B * f);
        const double dq = m1 + m2 + m3;
        const double q2 = -2.0 * (A + B + C);
        double disc = 4.0 * dq * dq + 6.0 * det * q2;
        if (disc < 0.0) disc = 0.0;
        const double ln = l + 3.0 * det / (dq + sqrt(disc));
        const bool adv = go && (dq > 0.0) && (ln > l);
        const double step = ln - l;
        if (adv) l = ln;
        go = adv && !(step <= tol);
    }
    const double A = a - l, B = b - l, C = c - l;
    const double m1 = B * C - e * e, m2 = A * C - f * f, m3 = A * B - d * d;
    const double am1 = fabs(m1), am2 = fabs(m2), am3 = fabs(m3);
    double x, y, z;
    if (am1 >= am2 && am1 >= am3) { x = m1; y = e * f - d * C; z = d * e - B * f; }
    else if (am2 >= am3) { x = e * f - d * C; y = m2; z = d * f - A * e; }
    else { x = d * e - B * f; y = d * f - A * e; z = m3; }
    const double n2 = x * x + y * y + z * z;
    if (n2 > 0.0) { const double inv = 1.0 / sqrt(n2); v[0] = x * inv; v[1] = y * inv; v[2] = z * inv; }
    else { v[0] = 0.0; v[1] = 0.0; v[2] = 1.0; }
    l0 = l;
}

// Stats::compute (AHCPlaneSeg.hpp:125-156); st = {sx,sy,sz,sxx,syy,szz,sxy,syz,sxz}
static __device__ void stats_compute_dev(const double *st, int N, double center[3], double normal[3], double &mse)
{
    const double sc = 1.0 / N;
    center[0] = st[0] * sc; center[1] = st[1] * sc; center[2] = st[2] * sc;
    const double k00 = st[3] - st[0] * st[0] * sc, k01 = st[6] - st[0] * st[1] * sc, k02 = st[8] - st[0] * st[2] * sc;
    const double k11 = st[4] - st[1] * st[1] * sc, k12 = st[7] - st[1] * st[2] * sc, k22 = st[5] - st[2] * st[2] * sc;
    double l0, v[3];
    eig33_smallest_dev(k00, k11, k22, k01, k12, k02, l0, v);
    if (v[0] * center[0] + v[1] * center[1] + v[2] * center[2] <= 0) {
        normal[0] = v[0]; normal[1] = v[1]; normal[2] = v[2];
    } else {
        normal[0] = -v[0]; normal[1] = -v[1]; normal[2] = -v[2];
    }
    mse = l0 * sc;
}

// ------------------------------------------------------------------------------------------------
// k_peac_blocks: one thread per 10x10 block
// ------------------------------------------------------------------------------------------------
// Node record of the grouped AHC kernel: everything an iteration reads of a node -- the sums and the normal (candidate
// evaluation), the header of its neighbour list, its disjoint set -- in ONE 128-byte line (segD + segI keep the blocks'
// fits, which the kernel reads once, and receive the extracted planes' records at the end).  A dead node has cnt == 0.
struct __attribute__((aligned(128))) HotNode { double st[9], nrm[3], mse; int N, rid, off, cnt, dss, dsr; };
static_assert(sizeof(HotNode) == 128, "HotNode is one cache line");
#ifdef HVO_WPE_BLOCKS
__attribute__((amdgpu_waves_per_eu(HVO_WPE_BLOCKS)))
#endif
__global__ __launch_bounds__(64) void k_peac_blocks(const uint16_t *__restrict__ depth, size_t dframe, int pitch,
                                                    int w, int h, int Nw, int nblk,
                                                    float fx, float fy, float cx, float cy, float dfac,
                                                    double *__restrict__ segD, int *__restrict__ segI, int segcap, HotNode *__restrict__ hot)
{
    const int blk = blockIdx.x * 64 + threadIdx.x, frame = blockIdx.y;
    if (blk >= nblk) return;
    const uint16_t *D = depth + (size_t)frame * dframe;
    const int bi = blk / Nw, bj = blk - bi * Nw;
    double st[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    int N = 0;
    bool valid = true;
    const double dfx = (double)fx, dfy = (double)fy, dcx = (double)cx, dcy = (double)cy, df = (double)dfac;
    const double rfx = 1.0 / dfx, rfy = 1.0 / dfy;                // hvo_div_const: 200 divisions by fx, fy per thread
    // A block row is 10 depths + the right neighbour of the last one: six aligned dwords (20 * bj bytes into a 64-byte
    // aligned row; the pair beyond the image edge is never used).  The row below is requested before this row is
    // processed and only needed for its "down" jump test, so its latency hides behind the row's arithmetic; the
    // reference's early exit at the first invalid pixel is a flag here (an invalid block keeps nothing of its sums).
    const int i0 = bi * WIN;
    const uint32_t *rp = reinterpret_cast<const uint32_t *>(D + (size_t)i0 * pitch + bj * WIN);
    const int rstep = pitch >> 1;                               // dwords per row
    uint32_t cur[6], nxt[6];
#pragma unroll
    for (int q = 0; q < 6; q++) cur[q] = rp[q];
    for (int ic = 0; ic < WIN && valid; ++ic) {
        const int i = i0 + ic;
        const bool has_down = i + 1 < h;
#pragma unroll
        for (int q = 0; q < 6; q++) nxt[q] = has_down ? rp[(size_t)(ic + 1) * rstep + q] : 0u;
        const double yk = (double)i - dcy;
        bool rowok = true;
        double zrow[WIN];
#pragma unroll
        for (int jc = 0; jc < WIN; ++jc) {
            const int j = bj * WIN + jc;
            const int d = (int)((cur[jc >> 1] >> (16 * (jc & 1))) & 0xFFFFu);
            rowok = rowok && d != 0;                            // ImagePointCloud::get: z == 0
            const double z = (double)d * df;
            zrow[jc] = z;
            if (j + 1 < w) {
                const int dn = (int)((cur[(jc + 1) >> 1] >> (16 * ((jc + 1) & 1))) & 0xFFFFu);
                if (dn != 0) { const double zn = (double)dn * df; if (fabs(z - zn) > 0.04 * fabs(z) + 0.02) rowok = false; }
            }
            const double x = hvo_div_const(((double)j - dcx) * z, dfx, rfx);
            const double y = hvo_div_const(yk * z, dfy, rfy);
            st[0] += x; st[1] += y; st[2] += z;
            st[3] += x * x; st[4] += y * y; st[5] += z * z;
            st[6] += x * y; st[7] += y * z; st[8] += x * z;
            ++N;
        }
        if (has_down) {
#pragma unroll
            for (int jc = 0; jc < WIN; ++jc) {
                const int dn = (int)((nxt[jc >> 1] >> (16 * (jc & 1))) & 0xFFFFu);
                if (dn != 0) { const double z = zrow[jc], zn = (double)dn * df; if (fabs(z - zn) > 0.04 * fabs(z) + 0.02) rowok = false; }
            }
        }
        valid = rowok;
#pragma unroll
        for (int q = 0; q < 6; q++) cur[q] = nxt[q];
    }
    double *sd = segD + ((size_t)frame * segcap + blk) * SEG_D;
    int *si = segI + ((size_t)frame * segcap + blk) * SEG_I;
    double center[3] = { 0, 0, 0 }, normal[3] = { 0, 0, 0 }, mse = 0;
    int ok = 0;
    if (valid && N >= 4) {
        stats_compute_dev(st, N, center, normal, mse);
        const double t = 1.6e-6 * center[2] * center[2] + 5.0;      // T_mse(P_INIT): AHCParamSet.hpp:88-100
        ok = mse < t * t;
    } else { N = 0; for (int k = 0; k < 9; k++) st[k] = 0; }
    for (int k = 0; k < 9; k++) sd[k] = st[k];
    sd[9] = center[0]; sd[10] = center[1]; sd[11] = center[2];
    sd[12] = normal[0]; sd[13] = normal[1]; sd[14] = normal[2]; sd[15] = mse;
    // [5], [7]: size and root of the node's disjoint set (a live node IS one set: DisjointSet::Union then needs no Find chain)
    si[0] = N; si[1] = blk; si[2] = ok ? 0 : 1; si[3] = blk * 4; si[4] = 0; si[5] = 1; si[6] = ok; si[7] = blk;
    // the grouped AHC kernel's record of the block (its list length is set there, with the edges)
    HotNode *hb = hot + (size_t)frame * segcap + blk;
    for (int k = 0; k < 9; k++) hb->st[k] = st[k];
    hb->nrm[0] = normal[0]; hb->nrm[1] = normal[1]; hb->nrm[2] = normal[2]; hb->mse = mse;
    hb->N = N; hb->rid = blk; hb->off = blk * 4; hb->cnt = 0; hb->dss = 1; hb->dsr = blk;
}

// ------------------------------------------------------------------------------------------------
// shared helpers for the single-wave kernels
// ------------------------------------------------------------------------------------------------
struct ClArgs {
    double *segD; int *segI; int *pool; int *pool2; int *parent; int *dsize; int *eflag; int *meta; int *extracted;
    struct HotNode *hot;                            // the grouped kernel's node records (one 128-byte line each)
    const int *perm;                                // wave b works on frames NG * perm[b] .. (hvo_frame_perm over the waves), or nullptr
    double *tqK, *tqM1k; int *tqM1i; int *tqLq; int tq_n0;      // the grouped kernel's min-MSE queue (TQueue): n0 * 256 keys, n0 * 16 bucket minima per frame
    int tq_lds_keys;                                 // GL = 64, a handful of frames: the keys and bucket minima live in LDS too (54 KB per frame)
    int segcap, poolcap, nblk, Nw, Nh;
    int edges_done;                                  // k_peac_edges has written eflag (initGraph's edges); the clustering kernels skip their own passes
    double c15, c60;
    double ang_factor, ang_near;   // T_ang(P_INIT): (angle_far - angle_near) / (z_far - z_near), angle_near (AHCParamSet.hpp:113-121)
};

// ParamSet::T_ang(P_INIT, z) (AHCParamSet.hpp:113-121).  With metric depth z never exceeds z_near = 500 and the threshold is
// the constant cos(15 deg) the host evaluated with its own libm (bit-identical to the oracle).  A depth_map_factor that
// yields z > 500 (e.g. DepthMapFactor 1 with depth in millimetres) takes the general branch; device cos() and glibc cos()
// may differ in the last bit there, which matters only for a block pair whose similarity equals the threshold to 1 ulp.
static __device__ __forceinline__ double t_ang_init(const ClArgs &a, double z)
{
    if (!(z > 500.0)) return a.c15;
    const double cz = z < 4000.0 ? z : 4000.0;
    return cos(a.ang_factor * cz + a.ang_near - a.ang_factor * 500.0);
}

static __device__ __forceinline__ double nsim(const double *a, const double *b)
{
    return fabs(a[12] * b[12] + a[13] * b[13] + a[14] * b[14]);
}

static __device__ int ds_find_ro(const int *parent, int x) { while (parent[x] != x) x = parent[x]; return x; }

// min-heap on (mse, id).  8-ary and wave-cooperative so that it can live in global memory (L2):
// the 8 children of a node are fetched by 8 lanes in one round trip, a 3-step shuffle reduction picks
// the smallest, i.e. a pop costs ~4 dependent accesses for 3072 entries.  Keeping the heap out of LDS
// is what lets ~16 frames per CU be resident (the kernel is latency bound, not LDS bound).
// Every call is made by all 64 lanes with uniform arguments; H.n is tracked uniformly.
struct Heap { double *key; int *id; int n; };
static __device__ __forceinline__ bool hless(double ka, int ia, double kb, int ib) { return ka < kb || (ka == kb && ia < ib); }
static __device__ void heap_sift_down(Heap &H, int i, double k, int id)
{
    const int lane = threadIdx.x & 63;
    for (;;) {
        const int c0 = 8 * i + 1;
        if (c0 >= H.n) break;
        double ck = 1.0e308; int cid = 0x7FFFFFFF, ci = -1;
        if (lane < 8 && c0 + lane < H.n) { ci = c0 + lane; ck = H.key[ci]; cid = H.id[ci]; }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const double ok = __shfl_xor(ck, o); const int oid = __shfl_xor(cid, o), oi = __shfl_xor(ci, o);
            if (oi >= 0 && (ci < 0 || hless(ok, oid, ck, cid))) { ck = ok; cid = oid; ci = oi; }
        }
        ck = __shfl(ck, 0); cid = __shfl(cid, 0); ci = __shfl(ci, 0);
        if (!hless(ck, cid, k, id)) break;
        if (lane == 0) { H.key[i] = ck; H.id[i] = cid; }
        i = ci;
    }
    if (lane == 0) { H.key[i] = k; H.id[i] = id; }
}
static __device__ void heap_push(Heap &H, double k, int id)
{
    const int lane = threadIdx.x & 63;
    int i = H.n++;
    while (i > 0) {
        const int p = (i - 1) / 8;
        const double pk = H.key[p]; const int pid = H.id[p];       // uniform load
        if (!hless(k, id, pk, pid)) break;
        if (lane == 0) { H.key[i] = pk; H.id[i] = pid; }
        i = p;
    }
    if (lane == 0) { H.key[i] = k; H.id[i] = id; }
}
// removes and returns the top id (uniform)
static __device__ int heap_pop(Heap &H)
{
    const int top = H.id[0];
    H.n--;
    if (H.n > 0) { const double k = H.key[H.n]; const int id = H.id[H.n]; __syncthreads(); heap_sift_down(H, 0, k, id); }
    __syncthreads();
    return top;
}

// remove up to two ids from a sorted neighbour list in place (single lane)
static __device__ void nb_remove2(int *pool, int *si, int a, int b)
{
    const int off = si[3], cnt = si[4];
    int o = 0;
    for (int k = 0; k < cnt; k++) { int v = pool[off + k]; if (v != a && v != b) { if (o != k) pool[off + o] = v; o++; } }
    si[4] = o;
}

// The core of ahCluster for one wave.  `heap` holds (mse,id); nodes are in segD/segI.
// Returns through meta/extracted.  Used for the main pass and for the last merge round.
// Compacts the live neighbour lists into the other pool buffer (ascending node id) and swaps the
// buffers.  The total live size never exceeds the initial 4*nblk entries (a merged list is at most
// |A|+|B|-2 long), so a pool of 16*nblk entries with compaction cannot run out.
static __device__ void pool_gc(int *segI, int nseg, int *&pool, int *&pool2, int &pooltop)
{
    const int lane = threadIdx.x;
    __syncthreads();
    int top = 0;
    for (int base = 0; base < nseg; base += 64) {
        const int id = base + lane;
        int sz = 0, off = 0;
        if (id < nseg) { const int *si = segI + (size_t)id * SEG_I; if (!si[2]) { sz = si[4]; off = si[3]; } }
        int incl = sz;
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        const int dst = top + incl - sz;
        for (int k = 0; k < sz; k++) pool2[dst + k] = pool[off + k];
        if (id < nseg) { int *si = segI + (size_t)id * SEG_I; if (!si[2]) { si[3] = dst; } }
        top += __shfl(incl, 63);
    }
    int *t = pool; pool = pool2; pool2 = t;
    pooltop = top;
    __syncthreads();
}

static __device__ void ah_cluster_wave(const ClArgs &a, int frame, Heap &H, int &nseg, int &pooltop, int *&pool, int *&pool2,
                                       int *ext, int &next, int *lA, int *lB, double *cm, int *cN, int &flags)
{
    const int lane = threadIdx.x;
    double *segD = a.segD + (size_t)frame * a.segcap * SEG_D;
    int *segI = a.segI + (size_t)frame * a.segcap * SEG_I;
    int *parent = a.parent + (size_t)frame * a.nblk, *dsize = a.dsize + (size_t)frame * a.nblk;
    while (H.n > 0) {
        if (pooltop > a.poolcap - 2 * a.nblk) pool_gc(segI, nseg, pool, pool2, pooltop);
        // ---- pop ----
        const int p = heap_pop(H);
        int *pi = segI + (size_t)p * SEG_I;
        if (pi[2]) continue;                                   // nouse
        const double *pd = segD + (size_t)p * SEG_D;
        const int pcnt = pi[4], poff = pi[3], pN = pi[0];
        // ---- evaluate merges with every neighbour, in creation order; one candidate per lane ----
        int cand_k = -1; double cand_mse = 0; int cand_N = 0;
        double st[9], c[3], n[3], m = 0; int mN = 0;           // the selected candidate (uniform after the loop)
        for (int base = 0; base < pcnt; base += 64) {
            const int k = base + lane;
            double lst[9], lc[3] = { 0, 0, 0 }, ln[3] = { 0, 0, 0 }, lm = 0; int lN = 0; bool has = false;
#pragma unroll
            for (int q = 0; q < 9; q++) lst[q] = 0;
            if (k < pcnt) {
                const int nb = pool[poff + k];
                const double *nd = segD + (size_t)nb * SEG_D;
                if (!(nsim(pd, nd) < a.c60)) {                 // T_ang(P_MERGING)
                    for (int q = 0; q < 9; q++) lst[q] = pd[q] + nd[q];
                    lN = pN + segI[(size_t)nb * SEG_I];
                    stats_compute_dev(lst, lN, lc, ln, lm);
                    has = true;
                }
            }
            cm[lane] = lm; cN[lane] = has ? lN : -1;
            __syncthreads();
            // exact sequential selection rule of AHCPlaneFitter.hpp:1043-1049 (evaluated uniformly)
            const int lim = min(64, pcnt - base);
            int sel = -1;
            for (int q = 0; q < lim; q++) {
                if (cN[q] < 0) continue;
                const double mq = cm[q];
                if (cand_k < 0 || cand_mse > mq || (cand_mse == mq && (double)cand_N < mq)) { cand_k = base + q; cand_mse = mq; cand_N = cN[q]; sel = q; }
            }
            if (sel >= 0) {                                     // take the winner's fit from its lane
#pragma unroll
                for (int q = 0; q < 9; q++) st[q] = __shfl(lst[q], sel);
#pragma unroll
                for (int q = 0; q < 3; q++) { c[q] = __shfl(lc[q], sel); n[q] = __shfl(ln[q], sel); }
                m = __shfl(lm, sel); mN = __shfl(lN, sel);
            }
            __syncthreads();
        }
        bool merged = false;
        if (cand_k >= 0) {
            const int nb = pool[poff + cand_k];
            int *ni = segI + (size_t)nb * SEG_I;
            const double t = 1.6e-6 * c[2] * c[2] + 8.0;        // T_mse(P_MERGING)
            if (m < t * t) {
                const int ncnt = ni[4], noff = ni[3];
                if (nseg >= a.segcap || pooltop + pcnt + ncnt > a.poolcap) {
                    flags |= 8;                                 // capacity: stop merging this node
                } else {
                    merged = true;
                    const int id = nseg++;
                    // new.nbs = (p.nbs U nb.nbs) \ {p, nb}: stage both lists in LDS when they fit (the
                    // usual case), lane 0 does the sorted merge; over-long lists are merged straight
                    // from global memory
                    const bool staged = pcnt <= LCAP && ncnt <= LCAP;
                    if (staged) {
                        for (int k = lane; k < pcnt; k += 64) lA[k] = pool[poff + k];
                        for (int k = lane; k < ncnt; k += 64) lB[k] = pool[noff + k];
                    }
                    __syncthreads();
                    const int *LA = staged ? lA : pool + poff, *LB = staged ? lB : pool + noff;
                    const int moff = pooltop;
                    int mcnt = 0;
                    if (lane == 0) {
                        int i = 0, j = 0;
                        while (i < pcnt || j < ncnt) {
                            int v;
                            if (j >= ncnt || (i < pcnt && LA[i] <= LB[j])) { v = LA[i]; if (j < ncnt && LB[j] == v) j++; i++; }
                            else { v = LB[j]; j++; }
                            if (v != p && v != nb) pool[moff + mcnt++] = v;
                        }
                        cN[0] = mcnt;
                        double *md = segD + (size_t)id * SEG_D;
                        for (int q = 0; q < 9; q++) md[q] = st[q];
                        md[9] = c[0]; md[10] = c[1]; md[11] = c[2]; md[12] = n[0]; md[13] = n[1]; md[14] = n[2]; md[15] = m;
                        int *mi = segI + (size_t)id * SEG_I;
                        mi[0] = mN; mi[1] = pN >= ni[0] ? pi[1] : ni[1]; mi[2] = 0; mi[3] = moff; mi[4] = mcnt; mi[5] = pcnt + ncnt; mi[6] = 1; mi[7] = 0;
                        // ds.Union(pa.rid, pb.rid)
                        int xr = ds_find_ro(parent, pi[1]), yr = ds_find_ro(parent, ni[1]);
                        if (xr != yr) {
                            if (dsize[xr] < dsize[yr]) { parent[xr] = yr; dsize[yr] += dsize[xr]; }
                            else { parent[yr] = xr; dsize[xr] += dsize[yr]; }
                        }
                        pi[2] = 1; ni[2] = 1; pi[4] = 0; ni[4] = 0;
                    }
                    __syncthreads();
                    heap_push(H, m, id);
                    __syncthreads();
                    mcnt = cN[0];
                    pooltop += pcnt + ncnt;
                    // every neighbour of the new node: drop p / nb, append the new id (largest so far)
                    for (int k = lane; k < mcnt; k += 64) {
                        int *qi = segI + (size_t)pool[moff + k] * SEG_I;
                        nb_remove2(pool, qi, p, nb);
                        pool[qi[3] + qi[4]] = id; qi[4]++;
                    }
                    __syncthreads();
                }
            }
        }
        if (!merged) {
            if (pN >= MIN_SUPPORT) { if (next < MAX_PLANES) { if (lane == 0) ext[next] = p; next++; } else flags |= 16; }
            // disconnectAllNbs
            for (int k = lane; k < pcnt; k += 64) nb_remove2(pool, segI + (size_t)pool[poff + k] * SEG_I, p, -1);
            __syncthreads();
            if (lane == 0) pi[4] = 0;
            __syncthreads();
        }
    }
    // std::sort by N descending, ties -> extraction order (stable insertion sort, uniform)
    __syncthreads();
    if (lane == 0) {
        for (int i = 1; i < next; i++) {
            int v = ext[i], j = i - 1;
            int vN = segI[(size_t)v * SEG_I];
            while (j >= 0 && segI[(size_t)ext[j] * SEG_I] < vN) { ext[j + 1] = ext[j]; j--; }
            ext[j + 1] = v;
        }
    }
    __syncthreads();
}

// ================================================================================================
// Grouped AHC: GL lanes per frame, 64/GL frames per wave, executed in lockstep.
// One frame keeps only a handful of lanes busy (a node has ~5 neighbours, and the 3x3 eigen-solve of
// the candidate merges dominates the instruction count), so four frames share a wave: every
// group-dependent loop becomes `while (any group still needs it)` with the body predicated per group,
// shuffles and ballots are confined to the group.  Semantics per frame are exactly those of
// ah_cluster_wave above (which remains in use, at one frame per wave, for the last merge round).
// ================================================================================================
#ifdef HVO_PEAC_TIMING
__device__ unsigned long long g_peac_t[32];
#define PT_DECL unsigned long long pt_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long pt_last = clock64();
#define PT(i) { const unsigned long long t_ = clock64(); pt_[i] += t_ - pt_last; pt_last = t_; }
#define PT_CNT(i, v) { pt_[i] += (v); }
#define PT_FLUSH if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 12; q_++) atomicAdd(&g_peac_t[q_], pt_[q_]); }
#else
#define PT_DECL
#define PT(i)
#define PT_CNT(i, v)
#define PT_FLUSH
#endif
// position(s) of `a` (and `b` when want2) in the unordered list pool[off .. off+cnt): the list is read
// eight entries per round trip (lists are short; a one-by-one scan would pay one memory latency per entry)
static __device__ __forceinline__ void list_find2(const int *pool, int off, int cnt, int a, int b, bool want2, int &i1, int &i2)
{
    i1 = -1; i2 = -1;
    for (int x0 = 0; x0 < cnt; x0 += 8) {
        int u[8];
#pragma unroll
        for (int j = 0; j < 8; j++) u[j] = x0 + j < cnt ? pool[off + x0 + j] : -1;
#pragma unroll
        for (int j = 0; j < 8; j++) if (u[j] == a || u[j] == b) { if (i1 < 0) i1 = x0 + j; else i2 = x0 + j; }
        if (i1 >= 0 && (!want2 || i2 >= 0)) break;
    }
}

// list_find2 in two halves, so that the first eight entries of several lists travel in one round trip
struct ListFind {
    int u[8], off, cnt;
    __device__ __forceinline__ void issue(const int *pool, int off_, int cnt_)
    {
        off = off_; cnt = cnt_;
#pragma unroll
        for (int j = 0; j < 8; j++) u[j] = j < cnt ? pool[off + j] : -1;
    }
    // the entries in registers only; true when the search is complete
    __device__ __forceinline__ bool scan(int a, int b, bool want2, int &i1, int &i2) const
    {
        i1 = -1; i2 = -1;
#pragma unroll
        for (int j = 0; j < 8; j++) if (u[j] >= 0 && (u[j] == a || u[j] == b)) { if (i1 < 0) i1 = j; else i2 = j; }
        return cnt <= 8 || (i1 >= 0 && (!want2 || i2 >= 0));
    }
    __device__ __forceinline__ void finish(const int *pool, int a, int b, bool want2, int &i1, int &i2) const
    {
        i1 = -1; i2 = -1;
#pragma unroll
        for (int j = 0; j < 8; j++) if (u[j] >= 0 && (u[j] == a || u[j] == b)) { if (i1 < 0) i1 = j; else i2 = j; }
        if (cnt > 8 && !(i1 >= 0 && (!want2 || i2 >= 0))) {
            for (int x0 = 8; x0 < cnt; x0 += 8) {
                int w[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w[j] = x0 + j < cnt ? pool[off + x0 + j] : -1;
#pragma unroll
                for (int j = 0; j < 8; j++) if (w[j] >= 0 && (w[j] == a || w[j] == b)) { if (i1 < 0) i1 = x0 + j; else i2 = x0 + j; }
                if (i1 >= 0 && (!want2 || i2 >= 0)) break;
            }
        }
    }
};

template <int GL> struct Grp {
    static_assert(GL == 16 || GL == 32 || GL == 64, "group width (a DPP row or more: TQueue reduces over 16-lane rows)");
    static __device__ __forceinline__ int gl() { return threadIdx.x & (GL - 1); }
    static __device__ __forceinline__ int gb() { return threadIdx.x & 63 & ~(GL - 1); }
    static __device__ __forceinline__ unsigned long long ballot(bool p)
    { const unsigned long long m = __ballot(p) >> gb(); return GL == 64 ? m : (m & ((1ull << GL) - 1)); }
    template <class T> static __device__ __forceinline__ T shfl(T v, int l) { return __shfl(v, gb() + l); }
};

static __device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Butterfly partner inside a 16-lane row through DPP (a VALU move) instead of a permute through LDS.  `old` = 0 with bound_ctrl: every lane
// of these permutations has its source inside the row, so the old value is never used -- and without it the compiler folds the DPP into the
// consuming v_min_i32 / v_max_i32 (one instruction per step instead of copy + move + op) and drops the copies in front of the 64-bit moves.
// STEP 1,2: quad_perm xor; STEP 4: row_half_mirror; STEP 8: row_mirror.  The mirrors pair lane i with
// 7-i / 15-i rather than i^4 / i^8, which is the same for an all-reduce of a commutative operation whose
// earlier steps made each quad / half-row uniform.  Steps >= 16 (wider groups) fall back to ds_bpermute.
template <int STEP> static __device__ __forceinline__ int row_partner(int v)
{
    if (STEP == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    if (STEP == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    if (STEP == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    if (STEP == 8) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return __shfl_xor(v, STEP);
}
template <int STEP> static __device__ __forceinline__ double row_partner(double v)
{
    return __hiloint2double(row_partner<STEP>(__double2hiint(v)), row_partner<STEP>(__double2loint(v)));
}

// ------------------------------------------------------------------------------------------------
// TQueue: the min-MSE queue of ahCluster for the grouped kernel, as a 16-ary TOURNAMENT over node ids.
//   K[id]            key of node id, +inf when it is not queued                       (global, 16 keys = one 128-byte line)
//   M1k/M1i[b]       smallest (key, id) of bucket b = ids 16 b .. 16 b + 15            (global)
//   M0k/M0i[s]       smallest (key, id) of the 16 buckets 16 s .. 16 s + 15             (LDS, n0 = ceil(segcap / 256) entries)
// The top is the minimum of M0 (n0 / 16 LDS reads per lane and one row reduction).  Changing the keys of up to three ids
// (a merge: p and its partner leave, the new node enters) re-reduces their buckets and super-buckets: nine independent line
// loads = ONE memory round trip, against one per level for a heap's sift; ids that share a bucket are patched in registers.
// Above all nothing is deleted lazily: with a heap the partner stayed queued until it was popped, and those dead pops
// were HALF of all iterations -- at four frames per wave in lockstep almost every one of them cost a full iteration.
// Ties: (key, id) lexicographic, as everywhere else (hless).
// ------------------------------------------------------------------------------------------------
#define TQ_INF 1.7976931348623157e308
struct TQueue { double *K, *M1k; int *M1i; double *M0k; int *M0i; int n0; };
// all-reduce of (k, i) under hless, key first: min of the keys through DPP (quad, half row, row, then the rows'
// last lanes broadcast onwards), then the smallest id among the lanes that hold that key -- 30 instructions against the ~80 of a
// reduction that carries (key, id) pairs through every step
static __device__ __forceinline__ double dpp_f64(double v, double old, const int ctrl_sel)
{
    int lo, hi;
    if (ctrl_sel == 0) { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true); }
    else if (ctrl_sel == 1) { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4E, 0xF, 0xF, true); }
    else if (ctrl_sel == 2) { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x141, 0xF, 0xF, true); }
    else if (ctrl_sel == 3) { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x140, 0xF, 0xF, true); }
    else if (ctrl_sel == 4) { lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x142, 0xA, 0xF, false); hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x142, 0xA, 0xF, false); }
    else { lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x143, 0xC, 0xF, false); hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x143, 0xC, 0xF, false); }
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ int dpp_i32(int v, int old, const int ctrl_sel)
{
    if (ctrl_sel == 0) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    if (ctrl_sel == 1) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    if (ctrl_sel == 2) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    if (ctrl_sel == 3) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    if (ctrl_sel == 4) return __builtin_amdgcn_update_dpp(old, v, 0x142, 0xA, 0xF, false);
    return __builtin_amdgcn_update_dpp(old, v, 0x143, 0xC, 0xF, false);
}
// min of two keys that are never NaN (an mse or TQ_INF): one v_min_f64 where `t < g ? t : g` costs a compare and two selects
static __device__ __forceinline__ double key_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int STEPS>                               // 4: within the 16-lane row (every lane of the row gets the result); 6: the wave (uniform result)
static __device__ __forceinline__ void min_key_id(double &k, int &i)
{
    double g = k;
#pragma unroll
    for (int s_ = 0; s_ < STEPS; s_++) { const double t = dpp_f64(g, g, s_); g = key_min(t, g); }
    if (STEPS == 6) g = readlane_f64(g, 63);
    int c = (k == g) ? i : 0x7FFFFFFF;
#pragma unroll
    for (int s_ = 0; s_ < STEPS; s_++) { const int t = dpp_i32(c, c, s_); c = min(c, t); }
    if (STEPS == 6) c = __builtin_amdgcn_readlane(c, 63);
    k = g; i = c;
}

// all-reduce of (k, i) under hless over the 16-lane DPP row (every lane ends up with the minimum)
static __device__ __forceinline__ void row_min16(double &k, int &i) { min_key_id<4>(k, i); }
// three independent reductions, their steps interleaved: a DPP instruction may not read a register the instruction before it wrote (two wait
// states), and one reduction is a chain of exactly such pairs -- side by side the three fill each other's gaps instead of s_nop
static __device__ __forceinline__ void row_min16x3(double (&k)[3], int (&i)[3])
{
    double g[3] = { k[0], k[1], k[2] };
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
        double t[3];
#pragma unroll
        for (int n = 0; n < 3; n++) t[n] = dpp_f64(g[n], g[n], s_);
#pragma unroll
        for (int n = 0; n < 3; n++) g[n] = key_min(t[n], g[n]);
    }
    int c[3];
#pragma unroll
    for (int n = 0; n < 3; n++) c[n] = (k[n] == g[n]) ? i[n] : 0x7FFFFFFF;
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
#pragma unroll
        for (int n = 0; n < 3; n++) { const int t = dpp_i32(c[n], c[n], s_); c[n] = min(c[n], t); }
    }
#pragma unroll
    for (int n = 0; n < 3; n++) { k[n] = g[n]; i[n] = c[n]; }
}
static __device__ __forceinline__ int tq_top(const TQueue &Q, int rl)
{
    double k = TQ_INF; int i = 0x7FFFFFFF;
    for (int j = rl; j < Q.n0; j += 16) { const double a = Q.M0k[j]; const int b = Q.M0i[j]; if (hless(a, b, k, i)) { k = a; i = b; } }
    row_min16(k, i);
    return k < TQ_INF ? i : -1;
}
// an update of nx (0, 1 or 3; uniform per group) keys, in two halves: issue() starts the loads, take() finishes and returns
// the new top.  rl = lane within the 16-lane row (wider groups compute every row redundantly), writer = the group's lane 0.
struct TQUpdate {
    double kk[3], mk[3], kn[3]; int mi[3], x[3], nx;
    __device__ __forceinline__ void issue(const TQueue &Q, int rl, int nx_, int x0, double k0, int x1, double k1, int x2, double k2)
    {
        nx = nx_; x[0] = x0; x[1] = x1; x[2] = x2; kn[0] = k0; kn[1] = k1; kn[2] = k2;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            kk[t] = TQ_INF; mk[t] = TQ_INF; mi[t] = 0x7FFFFFFF;
            if (t < nx) { kk[t] = Q.K[(x[t] & ~15) + rl]; const int j = ((x[t] >> 8) << 4) + rl; mk[t] = Q.M1k[j]; mi[t] = Q.M1i[j]; }
        }
    }
    __device__ __forceinline__ int take(const TQueue &Q, int rl, bool writer)
    {
#pragma unroll
        for (int t = 0; t < 3; t++) if (writer && t < nx) Q.K[x[t]] = kn[t];
        double bk[3]; int bi[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            double k = kk[t]; int i = (x[t] & ~15) + rl;
#pragma unroll
            for (int u = 0; u < 3; u++) if (u < nx && ((x[u] ^ x[t]) >> 4) == 0 && (x[u] & 15) == rl) k = kn[u];
            if (t >= nx) { k = TQ_INF; i = 0x7FFFFFFF; }
            bk[t] = k; bi[t] = i;
        }
        row_min16x3(bk, bi);
#pragma unroll
        for (int t = 0; t < 3; t++) if (writer && t < nx) { Q.M1k[x[t] >> 4] = bk[t]; Q.M1i[x[t] >> 4] = bi[t]; }
        double sk[3]; int si[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            double k = mk[t]; int i = mi[t];
#pragma unroll
            for (int u = 0; u < 3; u++) if (u < nx && ((x[u] ^ x[t]) >> 8) == 0 && ((x[u] >> 4) & 15) == rl) { k = bk[u]; i = bi[u]; }
            if (t >= nx) { k = TQ_INF; i = 0x7FFFFFFF; }
            sk[t] = k; si[t] = i;
        }
        row_min16x3(sk, si);
        double k = TQ_INF; int i = 0x7FFFFFFF;
        for (int j = rl; j < Q.n0; j += 16) {
            double a = Q.M0k[j]; int b = Q.M0i[j];
#pragma unroll
            for (int u = 0; u < 3; u++) if (u < nx && (x[u] >> 8) == j) { a = sk[u]; b = si[u]; }
            if (hless(a, b, k, i)) { k = a; i = b; }
        }
#pragma unroll
        for (int t = 0; t < 3; t++) if (writer && t < nx) { Q.M0k[x[t] >> 8] = sk[t]; Q.M0i[x[t] >> 8] = si[t]; }
        row_min16(k, i);
        return k < TQ_INF ? i : -1;
    }
};

// The same update for a frame that has the whole wave (GL = 64): the three positions are handled by three 16-lane rows side
// by side -- one bucket reduction, one super-bucket reduction and one top reduction instead of seven reductions in a row
// (a lone wave pays 5-8 cycles per instruction, and the queue update was a third of a single frame's iteration).
struct TQUpdateRows {
    double kk, mk, kn[3]; int mi, x[3], nx;
    __device__ __forceinline__ void issue(const TQueue &Q, int lane, int nx_, int x0, double k0, int x1, double k1, int x2, double k2)
    {
        nx = nx_; x[0] = x0; x[1] = x1; x[2] = x2; kn[0] = k0; kn[1] = k1; kn[2] = k2;
        const int row = lane >> 4, rl = lane & 15;
        const int xt = row == 0 ? x0 : row == 1 ? x1 : x2;
        kk = TQ_INF; mk = TQ_INF; mi = 0x7FFFFFFF;
        if (row < nx) { kk = Q.K[(xt & ~15) + rl]; const int j = ((xt >> 8) << 4) + rl; mk = Q.M1k[j]; mi = Q.M1i[j]; }
    }
    __device__ __forceinline__ int take(const TQueue &Q, int lane)
    {
        const int row = lane >> 4, rl = lane & 15;
        const int xt = row == 0 ? x[0] : row == 1 ? x[1] : x[2];
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < nx) Q.K[x[t]] = kn[t];
        }
        // my row's bucket
        double k = kk; int i = (xt & ~15) + rl;
#pragma unroll
        for (int u = 0; u < 3; u++) if (u < nx && ((x[u] ^ xt) >> 4) == 0 && (x[u] & 15) == rl) k = kn[u];
        if (row >= nx) { k = TQ_INF; i = 0x7FFFFFFF; }
        row_min16(k, i);
        double bk[3]; int bi[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            bk[u] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(k), 16 * u), __builtin_amdgcn_readlane(__double2loint(k), 16 * u));
            bi[u] = __builtin_amdgcn_readlane(i, 16 * u);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < nx) { Q.M1k[x[t] >> 4] = bk[t]; Q.M1i[x[t] >> 4] = bi[t]; }
        }
        // my row's super-bucket
        k = mk; i = mi;
#pragma unroll
        for (int u = 0; u < 3; u++) if (u < nx && ((x[u] ^ xt) >> 8) == 0 && ((x[u] >> 4) & 15) == rl) { k = bk[u]; i = bi[u]; }
        if (row >= nx) { k = TQ_INF; i = 0x7FFFFFFF; }
        row_min16(k, i);
        double sk[3]; int si[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            sk[u] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(k), 16 * u), __builtin_amdgcn_readlane(__double2loint(k), 16 * u));
            si[u] = __builtin_amdgcn_readlane(i, 16 * u);
        }
        // the top: 64 entries of M0 per step, a row reduction, then the four rows' results
        k = TQ_INF; i = 0x7FFFFFFF;
        for (int j = lane; j < Q.n0; j += 64) {
            double a = Q.M0k[j]; int b = Q.M0i[j];
#pragma unroll
            for (int u = 0; u < 3; u++) if (u < nx && (x[u] >> 8) == j) { a = sk[u]; b = si[u]; }
            if (hless(a, b, k, i)) { k = a; i = b; }
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < 3; t++) if (t < nx) { Q.M0k[x[t] >> 8] = sk[t]; Q.M0i[x[t] >> 8] = si[t]; }
        }
        row_min16(k, i);
        double tk = TQ_INF; int ti = 0x7FFFFFFF;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double rk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(k), 16 * r), __builtin_amdgcn_readlane(__double2loint(k), 16 * r));
            const int ri = __builtin_amdgcn_readlane(i, 16 * r);
            if (hless(rk, ri, tk, ti)) { tk = rk; ti = ri; }
        }
        return tk < TQ_INF ? ti : -1;
    }
};

template <int GL>
static __device__ void gpool_gc(HotNode *hot, int nseg, int *&pool, int *&pool2, int &pooltop, bool need)
{
    const int gl = Grp<GL>::gl();
    __syncthreads();
    int top = 0;
    for (int base = 0; __any(need && base < nseg); base += GL) {
        const int id = base + gl;
        int sz = 0, off = 0;
        const bool in = need && id < nseg;
        if (in) { sz = hot[id].cnt; off = hot[id].off; }                     // dead nodes: cnt == 0
        int incl = sz;
#pragma unroll
        for (int o = 1; o < GL; o <<= 1) { const int t = __shfl_up(incl, o, GL); if (gl >= o) incl += t; }
        const int dst = top + incl - sz;
        for (int k = 0; k < sz; k++) pool2[dst + k] = pool[off + k];
        if (in && sz) hot[id].off = dst;
        top += Grp<GL>::shfl(incl, GL - 1);
    }
    if (need) { int *t = pool; pool = pool2; pool2 = t; pooltop = top; }
    __syncthreads();
}

// Neighbour sets are kept UNORDERED here (the reference iterates a std::set<PlaneSeg*>, i.e. by
// address; the oracle fixes that as ascending creation id).  The only place where the iteration
// order can change the result is the candidate selection rule of AHCPlaneFitter.hpp:1043-1049
//     if (cand_mse > mse || (cand_mse == mse && cand->N < mse)) take
// walked in ascending id.  Restated without order: the winner has the minimum mse; among candidates
// that tie on it, walking in ascending id the cursor moves on while N(cursor) < mse, i.e. the winner
// is the smallest-id tied candidate with N >= mse, or the largest-id tied candidate if there is none.
// That is three reductions (min mse; min id with N >= mse; max id), so lists need no order, merging
// two lists is mark / test / compact in parallel, and removing an id is replace-or-swap-with-last.
#define ACH 1                // chunks (of GL neighbours) of the popped node whose list edits travel in one round trip
template <int GL>
static __device__ void ah_cluster_grouped(const ClArgs &a, int frame, const TQueue &Q, double *psl, int &hn, int &nseg, int &pooltop,
                                          int *&pool, int *&pool2, int *ext, int &next, int &flags)
{
    const int gl = Grp<GL>::gl(), gb = Grp<GL>::gb();
    HotNode *hot = a.hot + (size_t)frame * a.segcap;
    int *parent = a.parent + (size_t)frame * a.nblk, *dsize = a.dsize + (size_t)frame * a.nblk;
    const unsigned long long lt_mask = (1ull << gl) - 1;       // lanes of my group below me
    PT_DECL
    const int rl = gl & 15;
    __syncthreads();
    int ptop = hn > 0 ? tq_top(Q, rl) : -1;                     // hn = queued (= live) nodes
    // the popped node's record (sums, normal) is fetched one iteration ahead, underneath the list edits
    double nps[9], npn[3]; int n_cnt = 0, n_off = 0, n_N = 0, n_rid = 0, n_dsr = 0, n_dss = 0, a0n = -1;
    auto fetch_next = [&](int r) {
        const int q = r < 0 ? 0 : r;
        const HotNode *h = hot + q;
        n_cnt = h->cnt; n_off = h->off; n_N = h->N; n_rid = h->rid; n_dsr = h->dsr; n_dss = h->dss;
#pragma unroll
        for (int q2 = 0; q2 < 9; q2++) nps[q2] = h->st[q2];
        npn[0] = h->nrm[0]; npn[1] = h->nrm[1]; npn[2] = h->nrm[2];
    };
    fetch_next(ptop);
    a0n = gl < n_cnt ? pool[n_off + gl] : -1;
    while (__any(hn > 0)) {
        const bool act = hn > 0;
        PT_CNT(8, 1)
        const bool need_gc = act && pooltop > a.poolcap - 2 * a.nblk;
        if (__any(need_gc)) {
            gpool_gc<GL>(hot, nseg, pool, pool2, pooltop, need_gc);        // relocates the lists: fetch the list again
            n_off = hot[ptop < 0 ? 0 : ptop].off;
            a0n = gl < n_cnt ? pool[n_off + gl] : -1;
        }
        PT(0)
        const int p = act ? ptop : -1;
        HotNode *hp = hot + (p < 0 ? 0 : p);
        const bool live = act;                                 // every queued node is in use (nothing is deleted lazily)
        const int pcnt = live ? n_cnt : 0, poff = n_off, pN = n_N, prid = n_rid, pdsr = n_dsr, pdss = n_dss;
        // popped node: sums and normal, uniform per group; staged in LDS for the -DHVO_PEAC_PS_LDS form (see below)
        if (gl == 0) {
#pragma unroll
            for (int q = 0; q < 9; q++) psl[q] = nps[q];
            psl[9] = npn[0]; psl[10] = npn[1]; psl[11] = npn[2];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifndef HVO_PEAC_PS_LDS         // default: 24 registers per lane (56 ms alone); -DHVO_PEAC_PS_LDS: 12 doubles of LDS per group (157 registers, 60 ms alone,
                               // more room for the kernels beside it: worth 15 ms of the step under overlap policy 1, nothing under policy 5)
        double ps_r[9], pn_r[3];
#pragma unroll
        for (int q = 0; q < 9; q++) ps_r[q] = nps[q];
        pn_r[0] = npn[0]; pn_r[1] = npn[1]; pn_r[2] = npn[2];
        const double *ps = ps_r, *pn = pn_r;
#else
        const double *ps = psl, *pn = psl + 9;
#endif
        const int a0 = gl < pcnt ? a0n : -1;                   // first chunk of p's list, reused by the merge
        // ---- evaluate the merge with every neighbour, one candidate per lane; each lane keeps its best ----
        bool bhas = false; double bm = 0; int bid = 0x7FFFFFFF, bN = 0, gid = 0x7FFFFFFF, xid = -1;
        int brid = 0, bnoff = 0, bncnt = 0, bdsr = 0, bdss = 0; // the candidate's rid, set and list, fetched with its sums
        double bc[3] = { 0, 0, 0 }, bn[3] = { 0, 0, 0 };        // (the merged sums are re-formed when the record is written)
        PT(2)
        int id_cur = a0;                                        // ids of the next chunk are fetched one pass ahead
        int aid[ACH], aoff[ACH], acnt[ACH];                     // my neighbour of chunk u and its list header: pass A edits these lists
#pragma unroll
        for (int u = 0; u < ACH; u++) { aid[u] = -1; aoff[u] = 0; acnt[u] = 0; }
        for (int base = 0; __any(base < pcnt); base += GL) {
            PT_CNT(9, 1)
            const int k = base + gl;
            const int id_nxt = k + GL < pcnt ? pool[poff + k + GL] : -1;
            double lst[9]; int lN = 4, nb = 0, nrid = 0, noff_ = 0, ncnt_ = 0, ndsr = 0, ndss = 0; bool has = false;
#pragma unroll
            for (int q = 0; q < 9; q++) lst[q] = 0;
            if (k < pcnt) {
                nb = id_cur;
                const HotNode *nd = hot + nb;
                const int nN = nd->N; nrid = nd->rid; noff_ = nd->off; ncnt_ = nd->cnt; ndsr = nd->dsr; ndss = nd->dss;
                if (!(fabs(pn[0] * nd->nrm[0] + pn[1] * nd->nrm[1] + pn[2] * nd->nrm[2]) < a.c60)) {      // T_ang(P_MERGING)
#pragma unroll
                    for (int q = 0; q < 9; q++) lst[q] = ps[q] + nd->st[q];
                    lN = pN + nN;
                    has = true;
                }
#pragma unroll
                for (int u = 0; u < ACH; u++) if (base == u * GL) { aid[u] = nb; aoff[u] = noff_; acnt[u] = ncnt_; }
            }
            // one 3x3 eigen-solve pass serves every group (uniform call: no divergence inside)
            if (__any(has)) {
                double tc[3], tn[3], tm;
                PT(2)
                stats_compute_dev(lst, lN, tc, tn, tm);
                PT(4)
                if (has) {
                    const bool good = (double)lN >= tm, better = !bhas || tm < bm, equal = bhas && tm == bm;
                    if (better) { gid = good ? nb : 0x7FFFFFFF; xid = nb; }
                    else if (equal) { if (good && nb < gid) gid = nb; if (nb > xid) xid = nb; }
                    if (better || (equal && nb < bid)) {                   // payload follows (mse, id)
                        bid = nb; bN = lN; brid = nrid; bnoff = noff_; bncnt = ncnt_; bdsr = ndsr; bdss = ndss;
#pragma unroll
                        for (int q = 0; q < 3; q++) { bc[q] = tc[q]; bn[q] = tn[q]; }
                    }
                    if (better) bm = tm;
                    bhas = true;
                }
            }
            id_cur = id_nxt;
        }
        // ---- group reductions: min mse, then the tie rule ----
        double gmin = bhas ? bm : 1.7976931348623157e308;
#define GM_STEP(o) if (o < GL) { const double t = row_partner<o>(gmin); gmin = t < gmin ? t : gmin; }
        // a frame that owns the wave (GL = 64) rarely has more than one row of candidates: row 0's result is then the group's,
        // broadcast through a scalar instead of two more cross-row steps
        const bool one_row = GL == 64 && pcnt <= 16;
        GM_STEP(1) GM_STEP(2) GM_STEP(4) GM_STEP(8)
        if (one_row) gmin = readlane_f64(gmin, 0); else { GM_STEP(16) GM_STEP(32) }
#undef GM_STEP
        const bool tied = bhas && bm == gmin;
        int rg = tied ? gid : 0x7FFFFFFF, rx = tied ? xid : -1;
#define GM_STEP(o) if (o < GL) { const int t0 = row_partner<o>(rg), t1 = row_partner<o>(rx); rg = min(rg, t0); rx = max(rx, t1); }
        GM_STEP(1) GM_STEP(2) GM_STEP(4) GM_STEP(8)
        if (one_row) { rg = __builtin_amdgcn_readlane(rg, 0); rx = __builtin_amdgcn_readlane(rx, 0); } else { GM_STEP(16) GM_STEP(32) }
#undef GM_STEP
        const bool any_cand = Grp<GL>::ballot(bhas) != 0;
        const int win = rg != 0x7FFFFFFF ? rg : rx;            // neighbour id to merge with (if any_cand)
        // Only the quantities the group decides on are made uniform (mse, centre z, the partner's list); the
        // fit itself stays in the lane that computed it, and that lane writes the merged node's record.
        double m, c2; int noff, ncnt;
        bool is_w;                                             // this lane holds the winning fit in b*
        {
            const unsigned long long wm = Grp<GL>::ballot(tied && bid == win);
            const int wl = wm ? __ffsll((long long)wm) - 1 : 0, src = gb + wl;
            is_w = wm != 0 && gl == wl;
            m = __shfl(bm, src); c2 = __shfl(bc[2], src);
            noff = __shfl(bnoff, src); ncnt = __shfl(bncnt, src);
            // tie broken by the N-vs-mse clause towards a candidate whose fit no lane kept: fit it again
            // (uniformly over the group; lane 0 of the group then owns it)
            const bool refit = any_cand && wm == 0;
            if (__any(refit)) {
                double lst[9]; int lN = 4, rrid = 0, roff = 0, rcnt = 0, rdsr = 0, rdss = 0;
#pragma unroll
                for (int q = 0; q < 9; q++) lst[q] = 0;
                if (refit) {
                    const HotNode *nd = hot + win;
#pragma unroll
                    for (int q = 0; q < 9; q++) lst[q] = ps[q] + nd->st[q];
                    lN = pN + nd->N; rrid = nd->rid; roff = nd->off; rcnt = nd->cnt; rdsr = nd->dsr; rdss = nd->dss;
                }
                double tc[3], tn[3], tm;
                stats_compute_dev(lst, lN, tc, tn, tm);
                if (refit) {
#pragma unroll
                    for (int q = 0; q < 3; q++) { bc[q] = tc[q]; bn[q] = tn[q]; }
                    bm = tm; bN = lN; brid = rrid; bdsr = rdsr; bdss = rdss; bid = win; m = tm; c2 = tc[2]; noff = roff; ncnt = rcnt;
                    is_w = gl == 0;
                }
            }
        }
        PT(3)
        // ---- merge decision ----
        const int nb = win;
        bool do_merge = false;
        if (live && any_cand) {
            const double t = 1.6e-6 * c2 * c2 + 8.0;            // T_mse(P_MERGING)
            if (m < t * t) {
                if (nseg >= a.segcap || pooltop + pcnt + ncnt > a.poolcap) flags |= 8;     // capacity: keep the node unmerged
                else do_merge = true;
            }
        }
        if (!do_merge) ncnt = 0;
        const int id = nseg;
        const bool no_merge = live && !do_merge;
        if (do_merge) { PT_CNT(10, 1) nseg++; }
        if (no_merge) { PT_CNT(11, 1) }
        PT(4)
        // From here on an iteration is three independent chains of dependent loads: pass B (the partner's list -> its members'
        // list headers -> their lists), pass A (p's members' lists; their headers came with the evaluation) and the queue update.
        // They are issued together, stage by stage, so that a stage costs ONE memory round trip for all
        // three (loads of a wave return in order): 3-4 round trips instead of the 8-9 of running them one after the other.
        //   * new.nbs = (p.nbs U nb.nbs) \ {p, nb}, and every member's own list gets p / nb replaced by the new id.  Pass B walks
        //     nb's list: a member that also holds p is left to pass A (read only here); the others are copied and get nb -> id.
        //     Pass A walks p's list: copy, p -> id, nb dropped.  B only writes lists of nodes that are not p's neighbours and
        //     reads the others before A writes them (all of B, its tail included, precedes A's stores);
        //   * no merge: disconnectAllNbs = pass A with "drop p";
        //   * the queue (TQueue): p leaves; on a merge the partner leaves and the new node enters.  One round trip yields the
        //     next top, whose record is fetched underneath the remaining stages (everything in it but its list, which this
        //     iteration may still edit).
        const int moff = pooltop;
        if (act) hn -= 1;                                       // merge: -2 + 1
        TQUpdate qu; TQUpdateRows qr;                          // (GL = 64 uses the row-parallel form)
        // stage 0: the partner's first chunk, my member's list header, the queue's lines
        bool editA[ACH]; ListFind fa[ACH];                     // (the headers of p's members came with the evaluation's loads)
#pragma unroll
        for (int u = 0; u < ACH; u++) { editA[u] = aid[u] >= 0 && ((do_merge && aid[u] != nb) || no_merge); fa[u].issue(pool, aoff[u], editA[u] ? acnt[u] : 0); }
        const int vB = gl < ncnt ? pool[noff + gl] : -1;        // ncnt == 0 unless this group merges
        if (GL == 64) qr.issue(Q, gl, act ? (do_merge ? 3 : 1) : 0, p, TQ_INF, nb, TQ_INF, id, m);
        else qu.issue(Q, rl, act ? (do_merge ? 3 : 1) : 0, p, TQ_INF, nb, TQ_INF, id, m);
        // stage 1: B's members' list headers, A's lists, the next top's record
        const bool inB = vB >= 0 && vB != p;
        int qb_off = 0, qb_cnt = 0;
        if (inB) { qb_off = hot[vB].off; qb_cnt = hot[vB].cnt; }
        ptop = GL == 64 ? qr.take(Q, gl) : qu.take(Q, rl, gl == 0);
        if (hn <= 0) ptop = -1;
        PT(1)
        fetch_next(ptop);
        // stage 2: B's lists
        ListFind fb; fb.issue(pool, qb_off, inB ? qb_cnt : 0);
        int a_i1[ACH], a_i2[ACH];
        {
            // lists longer than the eight entries in registers: the chunks continue TOGETHER, eight more entries each per round trip
            bool more[ACH]; bool anymore = false;
#pragma unroll
            for (int u = 0; u < ACH; u++) { more[u] = !fa[u].scan(p, do_merge ? nb : p, do_merge, a_i1[u], a_i2[u]); anymore |= more[u]; }
            for (int x0 = 8; __any(anymore); x0 += 8) {
                int w[ACH][8];
#pragma unroll
                for (int u = 0; u < ACH; u++)
#pragma unroll
                    for (int j = 0; j < 8; j++) w[u][j] = (more[u] && x0 + j < fa[u].cnt) ? pool[fa[u].off + x0 + j] : -1;
                anymore = false;
#pragma unroll
                for (int u = 0; u < ACH; u++) {
                    const int fa_a = p, fa_b = do_merge ? nb : p;
#pragma unroll
                    for (int j = 0; j < 8; j++) if (w[u][j] >= 0 && (w[u][j] == fa_a || w[u][j] == fa_b)) { if (a_i1[u] < 0) a_i1[u] = x0 + j; else a_i2[u] = x0 + j; }
                    more[u] = more[u] && x0 + 8 < fa[u].cnt && !(a_i1[u] >= 0 && (!do_merge || a_i2[u] >= 0));
                    anymore |= more[u];
                }
            }
        }
        // stage 3: the edits.  All of B first ...
        int mcnt = 0;
        {
            int b_i1, b_i2;
            fb.finish(pool, nb, p, true, b_i1, b_i2);
            const bool keep = inB && b_i2 < 0;
            if (keep && b_i1 >= 0) pool[qb_off + b_i1] = id;
            const unsigned long long km = Grp<GL>::ballot(keep);
            if (keep) pool[moff + mcnt + __popcll(km & lt_mask)] = vB;
            mcnt += __popcll(km);
        }
        PT(5)
        for (int base = GL; __any(base < ncnt); base += GL) {    // partner lists longer than a chunk (rare)
            const int k = base + gl;
            int v = -1; bool keep = false;
            if (k < ncnt) {
                v = pool[noff + k];
                if (v != p) {
                    const int off = hot[v].off, cnt = hot[v].cnt;
                    int i1, i2;
                    list_find2(pool, off, cnt, nb, p, true, i1, i2);
                    if (i2 < 0) { keep = true; if (i1 >= 0) pool[off + i1] = id; }
                }
            }
            const unsigned long long km = Grp<GL>::ballot(keep);
            if (keep) pool[moff + mcnt + __popcll(km & lt_mask)] = v;
            mcnt += __popcll(km);
        }
        PT(6)
        // ... then A: the first of {p, nb} found becomes the new id, the second is dropped (merge); p is dropped (no merge)
#pragma unroll
        for (int u = 0; u < ACH; u++) {
            if (__any(aid[u] >= 0)) {
                if (editA[u]) {
                    HotNode *hv = hot + aid[u];
                    const int off = aoff[u], cnt = acnt[u], i1 = a_i1[u], i2 = a_i2[u];
                    if (do_merge) {
                        if (i1 >= 0) pool[off + i1] = id;
                        if (i2 >= 0) { if (i2 != cnt - 1) pool[off + i2] = pool[off + cnt - 1]; hv->cnt = cnt - 1; }
                    } else if (i1 >= 0) { if (i1 != cnt - 1) pool[off + i1] = pool[off + cnt - 1]; hv->cnt = cnt - 1; }
                }
                const bool keep = editA[u] && do_merge;
                const unsigned long long km = Grp<GL>::ballot(keep);
                if (keep) pool[moff + mcnt + __popcll(km & lt_mask)] = aid[u];
                mcnt += __popcll(km);
            }
        }
        PT(5)
        for (int base = ACH * GL; __any(live && base < pcnt); base += GL) {   // p's list beyond ACH chunks (rare)
            const int k = base + gl;
            int v = -1; bool keep = false;
            if (live && k < pcnt) {
                v = pool[poff + k];
                if (!do_merge || v != nb) {
                    HotNode *hv = hot + v;
                    const int off = hv->off, cnt = hv->cnt;
                    int i1, i2;
                    if (do_merge) {
                        keep = true;
                        list_find2(pool, off, cnt, p, nb, true, i1, i2);
                        if (i1 >= 0) pool[off + i1] = id;
                        if (i2 >= 0) { if (i2 != cnt - 1) pool[off + i2] = pool[off + cnt - 1]; hv->cnt = cnt - 1; }
                    } else {
                        list_find2(pool, off, cnt, p, p, false, i1, i2);
                        if (i1 >= 0) { if (i1 != cnt - 1) pool[off + i1] = pool[off + cnt - 1]; hv->cnt = cnt - 1; }
                    }
                }
            }
            const unsigned long long km = Grp<GL>::ballot(keep);
            if (keep) pool[moff + mcnt + __popcll(km & lt_mask)] = v;
            mcnt += __popcll(km);
        }
        PT(6)
        if (do_merge && is_w) {
            HotNode *M = hot + id;
            const HotNode *W = hot + bid;                                                 // sums of the merge = p's + the partner's
#pragma unroll
            for (int q = 0; q < 9; q++) M->st[q] = ps[q] + W->st[q];
            M->nrm[0] = bn[0]; M->nrm[1] = bn[1]; M->nrm[2] = bn[2]; M->mse = bm;
            // ds.Union(pa.rid, pb.rid) (DisjointSet.hpp:63-83): the two nodes carry their sets' roots and sizes -- two stores, no Find
            int root = pdsr, size = pdss + bdss;
            if (pdsr == bdsr) size = pdss;
            else if (pdss < bdss) { parent[pdsr] = bdsr; dsize[bdsr] = size; root = bdsr; }
            else { parent[bdsr] = pdsr; dsize[pdsr] = size; }
            M->N = bN; M->rid = pN >= bN - pN ? prid : brid; M->off = moff; M->cnt = mcnt; M->dss = size; M->dsr = root;
            hp->cnt = 0; hot[nb].cnt = 0;
        }
        if (do_merge) pooltop += pcnt + ncnt;
        if (no_merge) {
            if (pN >= MIN_SUPPORT) { if (next < MAX_PLANES) { if (gl == 0) ext[next] = p; next++; } else flags |= 16; }
            if (gl == 0) hp->cnt = 0;
        }
        PT(7)
        __syncthreads();                                       // this iteration's records, lists and queue stores
        // the next top's list (and all of its record when it is the node this iteration created)
        {
            const bool fresh = do_merge && ptop == id;
            if (__any(fresh)) { if (fresh) fetch_next(ptop); }
            n_cnt = hot[ptop < 0 ? 0 : ptop].cnt;
            if (fresh) { n_cnt = mcnt; n_off = moff; }
            a0n = gl < n_cnt ? pool[n_off + gl] : -1;
        }
        PT(0)
    }
    // std::sort by N descending, ties -> extraction order (stable insertion sort)
    __syncthreads();
    if (gl == 0) {
        for (int i = 1; i < next; i++) {
            int v = ext[i], j = i - 1;
            int vN = hot[v].N;
            while (j >= 0 && hot[ext[j]].N < vN) { ext[j + 1] = ext[j]; j--; }
            ext[j + 1] = v;
        }
    }
    __syncthreads();
    // what the refinement kernels read: segI of every node this kernel created says "dead, no list" (k_peac_final's pool
    // compaction walks them), and the extracted planes get their records (sums, centre, normal, mse; N, rid) in segD / segI
    {
        double *segD = a.segD + (size_t)frame * a.segcap * SEG_D;
        int *segI = a.segI + (size_t)frame * a.segcap * SEG_I;
        for (int v = a.nblk + gl; v < nseg; v += GL) { int *si = segI + (size_t)v * SEG_I; si[2] = 1; si[3] = 0; si[4] = 0; }
        __syncthreads();
        for (int e = gl; e < next; e += GL) {
            const int v = ext[e]; const HotNode *nv = hot + v;
            double *sd = segD + (size_t)v * SEG_D; int *si = segI + (size_t)v * SEG_I;
            const double sc = 1.0 / nv->N;
#pragma unroll
            for (int q = 0; q < 9; q++) sd[q] = nv->st[q];
            sd[9] = nv->st[0] * sc; sd[10] = nv->st[1] * sc; sd[11] = nv->st[2] * sc;
            sd[12] = nv->nrm[0]; sd[13] = nv->nrm[1]; sd[14] = nv->nrm[2]; sd[15] = nv->mse;
            si[0] = nv->N; si[1] = nv->rid; si[2] = 0; si[3] = 0; si[4] = 0; si[5] = 0; si[6] = 1; si[7] = 0;
        }
    }
    __syncthreads();
    PT(0)
    PT_FLUSH
}

// ------------------------------------------------------------------------------------------------
// k_peac_edges: initGraph's edges (AHCPlaneFitter.hpp:894-954) for the clustering kernels.  The two passes are state machines along
// a row / a column (the reference's loops step back and forth), 48 + 64 independent chains per 640x480 frame whose every step reads
// three or four block records: inside the clustering kernels (a quarter wave per frame, records in global memory) they were 7.5 % of
// the AHC's time.  Here a workgroup stages the frame's normals, centre depths and validity in LDS (34 bytes per block) and a thread
// walks a row, then a column, out of LDS.  Frames whose blocks do not fit (1280x960) keep the in-kernel passes (edges_done = 0).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_peac_edges(ClArgs a, int nframes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ed_lds[];
    const int frame = blockIdx.x, tid = threadIdx.x, nblk = a.nblk, Nw = a.Nw, Nh = a.Nh;
    double4 *nz = reinterpret_cast<double4 *>(ed_lds);                   // (normal, centre z) of a block
    unsigned char *okb = reinterpret_cast<unsigned char *>(nz + nblk), *ef = okb + nblk;
    const double *segD = a.segD + (size_t)frame * a.segcap * SEG_D;
    const int *segI = a.segI + (size_t)frame * a.segcap * SEG_I;
    int *eflag = a.eflag + (size_t)frame * nblk;
    for (int b = tid; b < nblk; b += 512) {
        const double *sd = segD + (size_t)b * SEG_D;
        nz[b] = make_double4(sd[12], sd[13], sd[14], sd[11]);
        okb[b] = segI[(size_t)b * SEG_I + 6] != 0; ef[b] = 0;
    }
    __syncthreads();
#define ENS(P_, Q_) fabs(nz[P_].x * nz[Q_].x + nz[P_].y * nz[Q_].y + nz[P_].z * nz[Q_].z)
    // rows (896-923).  bits: 1=left 2=right 4=up 8=down
    for (int i = tid; i < Nh; i += 512) {
        for (int j = 1; j < Nw; j += 2) {
            const int c = i * Nw + j;
            if (!okb[c - 1]) { --j; continue; }
            if (!okb[c]) continue;
            if (j < Nw - 1 && !okb[c + 1]) { ++j; continue; }
            const double th = t_ang_init(a, nz[c].w);
            if ((j < Nw - 1 && ENS(c - 1, c + 1) >= th) || (j == Nw - 1 && ENS(c, c - 1) >= th)) {
                ef[c] |= 1; ef[c - 1] |= 2;
                if (j < Nw - 1) { ef[c] |= 2; ef[c + 1] |= 1; }
            } else --j;
        }
    }
    __syncthreads();
    // columns (926-954)
    for (int j = tid; j < Nw; j += 512) {
        for (int i = 1; i < Nh; i += 2) {
            const int c = i * Nw + j;
            if (!okb[c - Nw]) { --i; continue; }
            if (!okb[c]) continue;
            if (i < Nh - 1 && !okb[c + Nw]) { ++i; continue; }
            const double th = t_ang_init(a, nz[c].w);
            if ((i < Nh - 1 && ENS(c - Nw, c + Nw) >= th) || (i == Nh - 1 && ENS(c, c - Nw) >= th)) {
                ef[c] |= 4; ef[c - Nw] |= 8;
                if (i < Nh - 1) { ef[c] |= 8; ef[c + Nw] |= 4; }
            } else --i;
        }
    }
#undef ENS
    __syncthreads();
    for (int b = tid; b < nblk; b += 512) eflag[b] = ef[b];
}

#include "peac_lend.inc"

// ------------------------------------------------------------------------------------------------
// k_peac_cluster: initGraph edges + main ahCluster, 64/GL frames per wave; LEND (GL = 16): idle lanes work for the other frames of the wave
// ------------------------------------------------------------------------------------------------
template <int GL, bool LEND>
#ifdef HVO_CLUSTER_WPE
__attribute__((amdgpu_waves_per_eu(HVO_CLUSTER_WPE, HVO_CLUSTER_WPE)))
#endif
__global__ __launch_bounds__(64) void k_peac_cluster(ClArgs a, int nframes)
{
    constexpr int NG = 64 / GL;
#ifdef HVO_CLUSTER_INFLATE      // experiment: allocate more registers without using them (tools/build_variant.sh)
    asm volatile("v_mov_b32 v231, 0" ::: "v231");
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char tq_lds[];          // NG groups x n0 x (double + int)
    __shared__ __attribute__((aligned(16))) unsigned char cl_lds[LEND ? sizeof(LendLds) : sizeof(double) * NG * 12];
    static_assert(!LEND || GL == 16, "lanes are lent between the four frames of a wave");
    const int lane = threadIdx.x, gl = Grp<GL>::gl(), gid = lane / GL;
    int frame = (a.perm ? a.perm[blockIdx.x] : (int)blockIdx.x) * NG + gid;
    const bool galive = frame < nframes;
    if (!galive) frame = nframes - 1;                          // idle group: aliases a frame read-only, writes nothing
    const int nblk = a.nblk, Nw = a.Nw, Nh = a.Nh;
    TQueue Q; Q.n0 = a.tq_n0;
    Q.K = a.tqK + (size_t)frame * Q.n0 * 256; Q.M1k = a.tqM1k + (size_t)frame * Q.n0 * 16; Q.M1i = a.tqM1i + (size_t)frame * Q.n0 * 16;
    Q.M0k = reinterpret_cast<double *>(tq_lds) + (size_t)gid * Q.n0;
    Q.M0i = reinterpret_cast<int *>(tq_lds + (size_t)NG * Q.n0 * sizeof(double)) + (size_t)gid * Q.n0;
    if (GL == 64 && a.tq_lds_keys) {
        // a lone frame owns its CU: the whole queue in LDS turns the update's memory round trip (a third of a single frame's iteration)
        // into LDS accesses
        unsigned char *kb = tq_lds + (((size_t)Q.n0 * 12 + 15) & ~(size_t)15);
        Q.K = reinterpret_cast<double *>(kb); Q.M1k = Q.K + (size_t)Q.n0 * 256; Q.M1i = reinterpret_cast<int *>(Q.M1k + (size_t)Q.n0 * 16);
    }
    double *segD = a.segD + (size_t)frame * a.segcap * SEG_D;
    int *segI = a.segI + (size_t)frame * a.segcap * SEG_I;
    int *pool = a.pool + (size_t)frame * a.poolcap, *pool2 = a.pool2 + (size_t)frame * a.poolcap;
    int *parent = a.parent + (size_t)frame * nblk, *dsize = a.dsize + (size_t)frame * nblk;
    int *eflag = a.eflag + (size_t)frame * nblk;
    HotNode *hot = a.hot + (size_t)frame * a.segcap;
#ifdef HVO_PEAC_TIMING
    const unsigned long long t_in0 = clock64();
#endif
    if (galive) for (int b = gl; b < nblk; b += GL) { parent[b] = b; dsize[b] = 1; if (!a.edges_done) eflag[b] = 0; }
    __syncthreads();
#define GOK(c) (segI[(size_t)(c) * SEG_I + 6] != 0)
#define SD(c) (segD + (size_t)(c) * SEG_D)
    // first pass: rows (AHCPlaneFitter.hpp:896-923).  bits: 1=left 2=right 4=up 8=down
    if (galive && !a.edges_done) for (int i = gl; i < Nh; i += GL) {
        for (int j = 1; j < Nw; j += 2) {
            const int c = i * Nw + j;
            if (!GOK(c - 1)) { --j; continue; }
            if (!GOK(c)) continue;
            if (j < Nw - 1 && !GOK(c + 1)) { ++j; continue; }
            const double th = t_ang_init(a, SD(c)[11]);   // T_ang(P_INIT, G[cidx]->center[2]), AHCPlaneFitter.hpp:903
            if ((j < Nw - 1 && nsim(SD(c - 1), SD(c + 1)) >= th) || (j == Nw - 1 && nsim(SD(c), SD(c - 1)) >= th)) {
                eflag[c] |= 1; eflag[c - 1] |= 2;
                if (j < Nw - 1) { eflag[c] |= 2; eflag[c + 1] |= 1; }
            } else --j;
        }
    }
    __syncthreads();
    if (galive && !a.edges_done) for (int j = gl; j < Nw; j += GL) {
        for (int i = 1; i < Nh; i += 2) {
            const int c = i * Nw + j;
            if (!GOK(c - Nw)) { --i; continue; }
            if (!GOK(c)) continue;
            if (i < Nh - 1 && !GOK(c + Nw)) { ++i; continue; }
            const double th = t_ang_init(a, SD(c)[11]);   // AHCPlaneFitter.hpp:933
            if ((i < Nh - 1 && nsim(SD(c - Nw), SD(c + Nw)) >= th) || (i == Nh - 1 && nsim(SD(c), SD(c - Nw)) >= th)) {
                eflag[c] |= 4; eflag[c - Nw] |= 8;
                if (i < Nh - 1) { eflag[c] |= 8; eflag[c + Nw] |= 4; }
            } else --i;
        }
    }
    __syncthreads();
    // adjacency lists in ascending id order (up, left, right, down); the queue's keys: a block's mse when it holds a node
    int hn = 0;
    for (int base = 0; base < Q.n0 * 256; base += GL) {
        const int b = base + gl;
        bool ok = false; double mse_b = 0;
        if (galive && b < nblk) {
            const int *si = segI + (size_t)b * SEG_I;
            const int e = eflag[b];
            int n = 0;
            if (e & 4) pool[b * 4 + n++] = b - Nw;
            if (e & 1) pool[b * 4 + n++] = b - 1;
            if (e & 2) pool[b * 4 + n++] = b + 1;
            if (e & 8) pool[b * 4 + n++] = b + Nw;
            ok = si[6] != 0;
            hot[b].cnt = n;                                      // (the rest of the record: k_peac_blocks)
            mse_b = hot[b].mse;
        }
        if (galive) Q.K[b] = ok ? mse_b : TQ_INF;
        hn += __popcll(Grp<GL>::ballot(ok));
    }
    __syncthreads();
#ifdef HVO_PEAC_TIMING
    const unsigned long long t_in1 = clock64();
#endif
    // the tournament above the keys: a lane reduces a whole bucket (16 contiguous keys), then a whole super-bucket
    if (galive) for (int j = gl; j < Q.n0 * 16; j += GL) {
        double k = TQ_INF; int i = 0x7FFFFFFF;
        for (int q = 0; q < 16; q++) { const double kq = Q.K[j * 16 + q]; if (hless(kq, j * 16 + q, k, i)) { k = kq; i = j * 16 + q; } }
        Q.M1k[j] = k; Q.M1i[j] = i;
    }
    __syncthreads();
    for (int j = gl; j < Q.n0; j += GL) {
        double k = TQ_INF; int i = 0x7FFFFFFF;
        if (galive) for (int q = 0; q < 16; q++) { const double kq = Q.M1k[j * 16 + q]; const int iq = Q.M1i[j * 16 + q]; if (hless(kq, iq, k, i)) { k = kq; i = iq; } }
        Q.M0k[j] = k; Q.M0i[j] = i;
    }
#ifdef HVO_PEAC_TIMING
    if (lane == 0) { const unsigned long long t_in2 = clock64(); atomicAdd(&g_peac_t[12], t_in1 - t_in0); atomicAdd(&g_peac_t[13], t_in2 - t_in1); atomicAdd(&g_peac_t[14], 1ull); }
#endif
    int nseg = nblk, pooltop = nblk * 4, next = 0, flags = 0;
    int *ext = a.extracted + (size_t)frame * 2 * MAX_PLANES;
    __syncthreads();
    if constexpr (LEND) ah_cluster_lend(a, frame, Q, *reinterpret_cast<LendLds *>(cl_lds), hn, nseg, pooltop, pool, pool2, ext, next, flags);
    else ah_cluster_grouped<GL>(a, frame, Q, reinterpret_cast<double *>(cl_lds) + gid * 12, hn, nseg, pooltop, pool, pool2, ext, next, flags);
    if (galive && gl == 0) {
        int *meta = a.meta + (size_t)frame * 16;
        meta[0] = nseg; meta[1] = 0; meta[2] = next; meta[3] = flags; meta[6] = 0;      // the lists are dead: k_peac_final starts an empty pool
    }
#undef GOK
#undef SD
}

#include "peac_heads.inc"
#include "peac_slots.inc"

// ------------------------------------------------------------------------------------------------
// k_peac_blkmap: findBlockMembership (block erosion) + coarse membership image
// ------------------------------------------------------------------------------------------------
// Per-pixel flood state, 4 bytes (round 5; 8 before), so that one scattered access per event fetches everything the
// floodFill state machine needs:  (int8 membership "trail") | FS_VALID (pixel of a block that
// survived the erosion: never touched) | depth << 16.  The reference's distMap entry is NOT stored: it is FLT_MAX until a pixel is
// claimed and from then on the point-plane distance of the pixel to the plane that owns it (trail >= 0) -- trail and distMap are only
// ever written together (AHCPlaneFitter.hpp:466-469) -- so the few events that contest an owned pixel recompute it from the owner's
// plane, with the operations that produced the stored value (k_peac_flood, `owner_dist`).  Half the bytes per state: half the lines a
// front touches, half of what k_peac_blkmap writes and k_peac_relabel reads, 1.2 MB per 640x480 frame less.
#define FS_VALID 0x100u
#define FS_LABEL(x) ((int)(signed char)((x) & 0xFFu))
// Flood-state layout: 8x4-pixel tiles, one 128-byte line each (a pixel's four neighbours mostly share its tile, and a front
// that advances through a tile finds it in L2 for several rings instead of one row-major line per ring): FS_TW tiles per row.
#define FS_TW(w) (((w) + 7) >> 3)
#define FS_TH(h) (((h) + 3) >> 2)
#define FS_IDX(x, y, tw) (((((y) >> 2) * (tw) + ((x) >> 3)) << 5) | (((y) & 3) << 3) | ((x) & 7))
#define FS_FRAME(w, h) ((size_t)FS_TW(w) * FS_TH(h) * 32)
#define FS_MAKE(lab, valid, d) (((unsigned)(lab) & 0xFFu) | ((valid) ? FS_VALID : 0u) | ((unsigned)(d) << 16))
__global__ __launch_bounds__(256) void k_peac_blkmap(const int *__restrict__ parent_, const int *__restrict__ dsize_,
                                                     const int *__restrict__ segI_, const int *__restrict__ ext_, const int *__restrict__ meta_,
                                                     int *__restrict__ blkmap_, int *__restrict__ isvalid_, uint32_t *__restrict__ state_,
                                                     const uint16_t *__restrict__ depth_, size_t dframe, int pitch,
                                                     int nblk, int Nw, int Nh, int w, int h, int segcap)
{
    // gridDim.y workgroups share a frame's pixels (a handful of frames: one workgroup filling 2.5 MB of states took 0.5 ms of the plane
    // chain); each of them computes the block map for itself -- the same values, written by all
    const int frame = blockIdx.x, tid = threadIdx.x, slice = blockIdx.y, nslices = gridDim.y;
    const int *parent = parent_ + (size_t)frame * nblk, *dsize = dsize_ + (size_t)frame * nblk;
    const int *segI = segI_ + (size_t)frame * segcap * SEG_I;
    const int *ext = ext_ + (size_t)frame * 2 * MAX_PLANES;
    const int next = meta_[(size_t)frame * 16 + 2];
    int *blkmap = blkmap_ + (size_t)frame * nblk, *isvalid = isvalid_ + (size_t)frame * MAX_PLANES;
    uint32_t *state = state_ + (size_t)frame * FS_FRAME(w, h);
    const uint16_t *D = depth_ + (size_t)frame * dframe;
    // phase 1: one thread per block
    for (int b = tid; b < nblk; b += 256) {
        const int i = b / Nw, j = b - i * Nw;
        const int setid = ds_find_ro(parent, b);
        int plid = -1;
        if (dsize[setid] * WIN * WIN >= MIN_SUPPORT) {
            bool same = true;
            if (j > 0 && ds_find_ro(parent, b - 1) != setid) same = false;
            if (same && j < Nw - 1 && ds_find_ro(parent, b + 1) != setid) same = false;
            if (same && i > 0 && ds_find_ro(parent, b - Nw) != setid) same = false;
            if (same && i < Nh - 1 && ds_find_ro(parent, b + Nw) != setid) same = false;
            if (same) {
                plid = 0;                                           // rid2plid[setid] (std::map default 0)
                for (int p = 0; p < next; p++) if (segI[(size_t)ext[p] * SEG_I + 1] == setid) plid = p;
                isvalid[plid] = 1;
            }
        }
        blkmap[b] = plid;
    }
    __syncthreads();
    // phase 2: the frame's pixels in state order (tile by tile: coalesced stores, depth read in 8-byte row pieces);
    // pixels outside the block grid get -1, tile padding outside the image is never read
    // a thread forms one row of a tile: eight states = 32 bytes (two 16-byte stores), their depths one 16-byte load (pitch and x are
    // multiples of 8; the frame's depth slab has a spare row behind it, and columns past the image are never read back); the row's
    // eight pixels lie in at most two blocks of the block grid
    const int tw = FS_TW(w), nrows = tw * FS_TH(h) * 4;
    for (int i = slice * 256 + tid; i < nrows; i += 256 * nslices) {
        const int t = i >> 2, ty = t / tw, tx = t - ty * tw;
        const int x0 = tx * 8, y = ty * 4 + (i & 3);
        if (y >= h) continue;
        const int by = y / WIN, bx0 = x0 / WIN, xs = (bx0 + 1) * WIN;            // pixels x >= xs belong to block column bx0 + 1
        const int l0 = (by < Nh && bx0 < Nw) ? blkmap[by * Nw + bx0] : -1, l1 = (by < Nh && bx0 + 1 < Nw) ? blkmap[by * Nw + bx0 + 1] : -1;
        unsigned dd[4];
        if (x0 + 8 <= pitch) { const uint4 q = *reinterpret_cast<const uint4 *>(D + (size_t)y * pitch + x0); dd[0] = q.x; dd[1] = q.y; dd[2] = q.z; dd[3] = q.w; }
        else { for (int k = 0; k < 4; k++) { const int xa = x0 + 2 * k; dd[k] = (xa < w ? (unsigned)D[(size_t)y * pitch + xa] : 0u) | ((xa + 1 < w ? (unsigned)D[(size_t)y * pitch + xa + 1] : 0u) << 16); } }
        unsigned st[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int lab = x0 + k < xs ? l0 : l1;
            st[k] = FS_MAKE(lab, lab >= 0, (dd[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(state + (size_t)i * 8);
        dst[0] = make_uint4(st[0], st[1], st[2], st[3]); dst[1] = make_uint4(st[4], st[5], st[6], st[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// k_peac_refine: seeds, floodFill, last merge round, plidmap.  One wave per frame.
// ------------------------------------------------------------------------------------------------
struct RfArgs {
    ClArgs c;
    const uint16_t *depth; size_t dframe; int pitch, w, h;
    float fx, fy, cx, cy, dfac;
    int *blkmap; int *isvalid; uint32_t *state; int *queue; int qcap; int *plidmap; const int *perm;
    hvo_plane *planes; double c30;
};

// k_peac_flood: seeds + floodFill (AHCPlaneFitter.hpp:543-575, 428-476), FLOOD_T threads per frame.
// Events = (queue entry, neighbour) in queue order.  A round takes the next FLOOD_T queue ENTRIES that
// existed when it started; a thread owns one entry and fetches the states of its (up to) four
// neighbour pixels, the only scattered reads.  Two thirds of the events are no-ops that the state
// fetched at the start of the round already proves to be no-ops:
//   * trail <= -6 (given up) never changes again;
//   * trail == plane ("passive"): the pixel already belongs to the event's plane with
//     dist == cdist(plane, pixel).  Whatever other planes do to the pixel earlier in the same round,
//     the event cannot win it back (their cdist is smaller), and its only possible side effect, the
//     adjacency bit (label-at-that-time, plane), duplicates the bit the displacing event set itself -
//     unless TWO different other planes changed the pixel before it in the same round.
// The remaining "live" events are compacted in event order into LDS records and processed
// ceil(live / FLOOD_T) at a time, so the expensive part (point-plane distance in double, grouping,
// state machine, append) is issued for dense waves only.  An LDS hash groups the live events by
// pixel.  If all live events of a pixel carry the same plane (the rule), the group has a closed form:
// the first event either wins the pixel (the others then see trail == plane) or every event of the
// group decrements the negative trail once (clamped at -6) / leaves a foreign label alone; only the
// first event acts.  A round in which some pixel collects live events of two different planes, or
// more than four live events, is "complex" (the passive events could matter, ranks would): it is
// replayed by one thread in plain queue order, exactly like the reference loop.  Pushes are appended
// in event order with a block scan, which reproduces the reference's queue order exactly.
// Queue entries are packed plid<<26 | y<<13 | x.
#ifndef FLOOD_HS_MUL
#define FLOOD_HS_MUL 1      // hash slots per event
#endif
#define FQ_PACK(x, y, pl) (((pl) << 26) | ((y) << 13) | (x))
// workgroup barrier that orders LDS traffic only: outstanding global stores are not waited for
static __device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int FLOOD_T, int EPL>
#ifdef HVO_WPE_FLOOD
__attribute__((amdgpu_waves_per_eu(HVO_WPE_FLOOD)))
#endif
__global__ __launch_bounds__(FLOOD_T) void k_peac_flood(RfArgs r, unsigned long long *__restrict__ adj_out)
{
    constexpr int NENT = FLOOD_T * EPL, NEV = NENT * 4;                        // queue entries / events per round (EPL entries per thread)
    constexpr int FLOOD_HS = NEV * FLOOD_HS_MUL, FLOOD_HL = 4, FLOOD_NP = 4 * EPL;   // FLOOD_NP = passes that cover all NEV events
    constexpr int NW = FLOOD_T / 64;
    static_assert(NEV <= 2048, "the group lists hold 11-bit event indices");
    constexpr int IXB = NEV > 1024 ? 11 : 10;                     // bits of a compact event index; (plane << IXB) | index is 16 bits, or 17 in a 32-bit entry
    constexpr unsigned IXM = (1u << IXB) - 1u;
    // group lists: (plane << IXB) | compact index per entry; the one-wave form (256 events) stores the index alone, in a byte, and reads the plane
    // from the event's record -- 1 KB instead of 2, which with the 8-byte records makes 16 workgroups per CU instead of 12 (the kernel is a chain of
    // dependent instructions per wave: ~1000 of every kind per round at ~1.8 ns each, profiles/r05_valu_salu_issue.txt; frames in flight are its throughput)
    constexpr bool HL8 = NEV <= 256;
    typedef typename std::conditional<HL8, unsigned char, typename std::conditional<(NEV > 1024), unsigned int, unsigned short>::type>::type hl_t;
    __shared__ double pl[MAX_PLANES][7];          // center[3], normal[3], mse
    __shared__ unsigned long long adj[MAX_PLANES], simok[MAX_PLANES];
    __shared__ int hkeys[FLOOD_HS], hcnt[FLOOD_HS];       // (a ranked round borrows both, by compact event index, once they are empty again: distances and push flags)
    __shared__ __attribute__((aligned(16))) hl_t hlist[FLOOD_HS * FLOOD_HL];
    __shared__ uint2 rec[NEV];                    // the round's live events in event order: packed (plane, y, x) of the target pixel, its state word
    static_assert(FLOOD_HS >= NEV, "a ranked round indexes hkeys / hcnt by compact event index");
    __shared__ int wsum[EPL * NW], psum[FLOOD_NP * NW];
    __shared__ int s_nq, s_cx[2];
    const ClArgs &a = r.c;
    const int frame = r.perm ? r.perm[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;     // hvo_frame_perm
    const int w = r.w, h = r.h, Nw = a.Nw, nblk = a.nblk;
    int *meta = a.meta + (size_t)frame * 16;
    const int nold = meta[2];
    const int *ext = a.extracted + (size_t)frame * 2 * MAX_PLANES;
    const double *segD = a.segD + (size_t)frame * a.segcap * SEG_D;
    const int *blkmap = r.blkmap + (size_t)frame * nblk;
    uint32_t *state = r.state + (size_t)frame * FS_FRAME(w, h);
    const int stw = FS_TW(w);
    int *queue = r.queue + (size_t)frame * r.qcap;
    int flags = 0;
    if (tid < MAX_PLANES) {
        adj[tid] = 0;
        if (tid < nold) { const double *sd = segD + (size_t)ext[tid] * SEG_D; for (int k = 0; k < 7; k++) pl[tid][k] = sd[9 + k]; }
    }
    for (int i = tid; i < FLOOD_HS; i += FLOOD_T) { hkeys[i] = -1; hcnt[i] = 0; }
    if (tid < 2) s_cx[tid] = 0;
    __syncthreads();
    // which pairs of planes count as adjacent when they meet on a pixel (|n_p . n_q| >= cos 30 deg,
    // AHCPlaneFitter.hpp:457-462): evaluated once per pair instead of once per contested event
    if (tid < MAX_PLANES) {
        unsigned long long m = 0;
        if (tid < nold) for (int q = 0; q < nold; q++) {
            const double *P = pl[tid], *Q = pl[q];
            if (fabs(P[3] * Q[3] + P[4] * Q[4] + P[5] * Q[5]) >= r.c30) m |= 1ull << q;
        }
        simok[tid] = m;
    }
    // ---- seeds in block raster order: count, scan, write.  One wave per frame walks the blocks 64 at a time; a workgroup of several waves
    //      (small batches) deals the groups of 64 out, counts first, and every wave places its groups behind the totals of the groups before ----
    constexpr int SEED_MAXG = 256;                           // groups of 64 blocks the parallel form holds (16384 blocks)
    __shared__ int gtot[NW > 1 ? SEED_MAXG : 1];
    auto seed_count = [&](int b, int &m, int &up, int &lf, int &i, int &j) {
        int cnt = 0; m = -2; up = -2; lf = -2; i = 0; j = 0;
        if (b < nblk) {
            i = b / Nw; j = b - i * Nw; m = blkmap[b];
            up = i > 0 ? blkmap[b - Nw] : -2; lf = j > 0 ? blkmap[b - 1] : -2;
            if (m < 0) { if (i > 0 && up >= 0) cnt += WIN - 1; if (j > 0 && lf >= 0) cnt += WIN - 1; }
            else { if (i > 0 && up != m) cnt += WIN - 1; if (j > 0 && lf != m) cnt += WIN - 1; }
        }
        return cnt;
    };
    auto seed_write = [&](int pos, int cnt, int m, int up, int lf, int i, int j) {
        if (cnt && pos + cnt <= r.qcap) {
            const int x0 = j * WIN, y0 = i * WIN;
            if (m < 0) {
                if (i > 0 && up >= 0) for (int k = 1; k < WIN; ++k) queue[pos++] = FQ_PACK(x0 + k, y0 - 1, up);
                if (j > 0 && lf >= 0) for (int k = 0; k < WIN - 1; ++k) queue[pos++] = FQ_PACK(x0 - 1, y0 + k, lf);
            } else {
                if (i > 0 && up != m) for (int k = 0; k < WIN - 1; ++k) queue[pos++] = FQ_PACK(x0 + k, y0, m);
                if (j > 0 && lf != m) for (int k = 1; k < WIN; ++k) queue[pos++] = FQ_PACK(x0, y0 + k, m);
            }
        }
    };
    const int ngrp = (nblk + 63) / 64;
    if (NW > 1 && ngrp <= SEED_MAXG) {
        for (int g = wv; g < ngrp; g += NW) {
            int m, up, lf, i, j;
            int incl = seed_count(g * 64 + lane, m, up, lf, i, j);
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            if (lane == 63) gtot[g] = incl;
        }
        __syncthreads();
        int before = 0, gdone = 0;                           // records of the groups before g (this wave's groups come in ascending order)
        for (int g = wv; g < ngrp; g += NW) {
            for (; gdone < g; gdone++) before += gtot[gdone];
            int m, up, lf, i, j;
            const int cnt = seed_count(g * 64 + lane, m, up, lf, i, j);
            int incl = cnt;
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            seed_write(before + incl - cnt, cnt, m, up, lf, i, j);
        }
        if (tid == 0) { int t = 0; for (int g = 0; g < ngrp; g++) t += gtot[g]; s_nq = t; }
    } else if (wv == 0) {
        int nq = 0;
        for (int base = 0; base < nblk; base += 64) {
            int m, up, lf, i, j;
            const int cnt = seed_count(base + lane, m, up, lf, i, j);
            int incl = cnt;
            for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            seed_write(nq + incl - cnt, cnt, m, up, lf, i, j);
            nq += __shfl(incl, 63);
        }
        if (lane == 0) s_nq = nq;
    }
    __syncthreads();
    int nq = s_nq;
    if (nq > r.qcap) { nq = r.qcap; flags |= 32; }
    const double dfx = (double)r.fx, dfy = (double)r.fy, dcx = (double)r.cx, dcy = (double)r.cy, df = (double)r.dfac;
    const double rfx = 1.0 / dfx, rfy = 1.0 / dfy;                // hvo_div_const
#ifdef HVO_PEAC_TIMING
    unsigned long long ft[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, ftl = clock64();
#define FT(i) { const unsigned long long t_ = clock64(); ft[i] += t_ - ftl; ftl = t_; }
#else
#define FT(i)
#endif
    FT(0)
    // point-plane distance of pixel (px, py) with raw depth d to coarse plane P (AHCPlaneFitter.hpp:463-467)
    // `own` >= 0: the plane that owns the pixel; odist = its distMap entry, recomputed (FLT_MAX for a pixel nobody has claimed).  An owned
    // pixel has a depth (only an event that passed the distance test claims), so d != 0 whenever own >= 0.
    auto geom = [&](const double *P, int px, int py, int d, int own, float &cdist, bool &ok, float &odist) {
        cdist = -1; ok = false; odist = __uint_as_float(0x7F7FFFFFu);
        if (d != 0) {
            const double z = (double)d * df;
            const double x = hvo_div_const(((double)px - dcx) * z, dfx, rfx), y = hvo_div_const(((double)py - dcy) * z, dfy, rfy);
            const double sd = P[3] * (x - P[0]) + P[4] * (y - P[1]) + P[5] * (z - P[2]);
            cdist = (float)fabs(sd);
            ok = ((double)cdist * (double)cdist) < 9 * P[6] + 1e-5;
            if (own >= 0) {
                const double *O = pl[own];
                odist = (float)fabs(O[3] * (x - O[0]) + O[4] * (y - O[1]) + O[5] * (z - O[2]));
            }
        }
    };
    const unsigned long long ltm = (1ull << lane) - 1;
    int kq = 0, par = 0, pf_nq = 0;                      // entry cursor; entries prefetched for the next round, valid for k < pf_nq
    int n_rounds = 0, n_ranked = 0, n_serial = 0;        // diagnostics (hvo_debug_peac_stats)
    int qpf[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) qpf[e] = 0;
    while (kq < nq) {
#ifdef HVO_PEAC_TIMING
        ft[4]++;
#endif
        const int nent = min(NENT, nq - kq);
        n_rounds++;
        if (tid == 0) s_cx[par ^ 1] = 0;
        int q[EPL]; bool own[EPL];                       // thread tid owns entries kq + e * FLOOD_T + tid
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int i = e * FLOOD_T + tid, k = kq + i;
            own[e] = i < nent;
            q[e] = qpf[e];
            if (own[e] && k >= pf_nq) q[e] = queue[k];
        }
        if (nent == NENT) {
#pragma unroll
            for (int e = 0; e < EPL; e++) { const int k = kq + NENT + e * FLOOD_T + tid; if (k < nq) qpf[e] = queue[k]; }
            pf_nq = nq;
        } else pf_nq = 0;
        // ---- fetch the four neighbour states (getValid4Neighbor order: left, right, up, down), keep the live events ----
        int wtot = 0;
        {
            uint32_t st[EPL][4]; int below[EPL], etot[EPL], off[EPL];
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int sx = q[e] & 8191, sy = (q[e] >> 13) & 8191;
                const bool ex[4] = { own[e] && sx > 0, own[e] && sx < w - 1, own[e] && sy > 0, own[e] && sy < h - 1 };
                const int px[4] = { sx - 1, sx + 1, sx, sx }, py[4] = { sy, sy, sy - 1, sy + 1 };
#pragma unroll
                for (int j = 0; j < 4; j++) { st[e][j] = FS_VALID; if (ex[j]) st[e][j] = state[FS_IDX(px[j], py[j], stw)]; }   // the only scattered reads
            }
            unsigned actm = 0;
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int plid = (int)((unsigned)q[e] >> 26);
                below[e] = 0; etot[e] = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int trail = FS_LABEL(st[e][j]);
                    const bool act = !(st[e][j] & FS_VALID) && trail > -6 && trail != plid;     // pixels of still-valid blocks are never touched
                    if (act) actm |= 1u << (e * 4 + j);
                    const unsigned long long bm = __ballot(act);
                    below[e] += __popcll(bm & ltm); etot[e] += __popcll(bm);
                }
            }
            // compact index = rank in event order (entry, neighbour): entries e * FLOOD_T + tid, so (e, wave, lane, j)
            if (NW > 1) {
                if (lane == 0) for (int e = 0; e < EPL; e++) wsum[e * NW + wv] = etot[e];
                lds_barrier();
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    off[e] = wtot;
                    for (int i = 0; i < NW; i++) { const int v = wsum[e * NW + i]; if (i < wv) off[e] += v; wtot += v; }
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; e++) { off[e] = wtot; wtot += etot[e]; }
            }
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const int plid = (int)((unsigned)q[e] >> 26), sx = q[e] & 8191, sy = (q[e] >> 13) & 8191;
                const int px[4] = { sx - 1, sx + 1, sx, sx }, py[4] = { sy, sy, sy - 1, sy + 1 };
                int c = off[e] + below[e];
#pragma unroll
                for (int j = 0; j < 4; j++) if (actm & (1u << (e * 4 + j))) { rec[c] = make_uint2((unsigned)FQ_PACK(px[j], py[j], plid), st[e][j]); c++; }
            }
        }
        lds_barrier();
        const int na = __builtin_amdgcn_readfirstlane(wtot);
        const int npass = (na + FLOOD_T - 1) / FLOOD_T;
#ifdef HVO_PEAC_TIMING
        if (tid == 0) { atomicAdd(&g_peac_t[30], (unsigned long long)na); atomicAdd(&g_peac_t[31], (unsigned long long)npass); }
#endif
        FT(1)
        // ---- live events, FLOOD_T at a time: distance, then group by pixel ----
        unsigned eq[FLOOD_NP], es[FLOOD_NP]; float ed[FLOOD_NP], ecd[FLOOD_NP]; bool eok[FLOOD_NP], ev[FLOOD_NP], push[FLOOD_NP]; int hs[FLOOD_NP], eix[FLOOD_NP];
#pragma unroll
        for (int p = 0; p < FLOOD_NP; p++) {
            ev[p] = false; push[p] = false; eq[p] = 0; es[p] = 0; ed[p] = 0; ecd[p] = -1; eok[p] = false; hs[p] = 0; eix[p] = 0;
            if (p < npass) {
                const int c = p * FLOOD_T + tid;
                ev[p] = c < na;
                if (ev[p]) {
                    const uint2 R = rec[c];
                    eq[p] = R.x; es[p] = R.y;
                    const int ep = (int)(R.x >> 26), ex_ = (int)(R.x & 8191u), ey = (int)((R.x >> 13) & 8191u);
                    eix[p] = FS_IDX(ex_, ey, stw);
                    geom(pl[ep], ex_, ey, (int)(R.y >> 16), FS_LABEL(R.y), ecd[p], eok[p], ed[p]);
                    int s = (int)(((unsigned)eix[p] * 2654435761u) >> 19) & (FLOOD_HS - 1);
                    for (;;) {
                        const int old = atomicCAS(&hkeys[s], -1, eix[p]);
                        if (old == -1 || old == eix[p]) break;
                        s = (s + 1) & (FLOOD_HS - 1);
                    }
                    const int pos = atomicAdd(&hcnt[s], 1);
                    if (pos < FLOOD_HL) hlist[s * FLOOD_HL + pos] = HL8 ? (hl_t)c : (hl_t)((ep << IXB) | c);
                    hs[p] = s;
                }
            }
        }
        lds_barrier();
        // ---- first event of every pixel group, group size; groups that have no closed form flag the round ----
        bool first[FLOOD_NP], multi[FLOOD_NP]; int cnt[FLOOD_NP]; int cx = 0;
#pragma unroll
        for (int p = 0; p < FLOOD_NP; p++) {
            first[p] = false; multi[p] = false; cnt[p] = 1;
            if (p < npass && ev[p]) {
                cnt[p] = hcnt[hs[p]];
                first[p] = true;
                if (cnt[p] > FLOOD_HL) cx |= 2;
                else if (cnt[p] > 1) {
                    const unsigned me = ((eq[p] >> 26) << IXB) | (unsigned)(p * FLOOD_T + tid);
                    const hl_t *hp_ = &hlist[hs[p] * FLOOD_HL];
                    unsigned l[4] = { hp_[0], hp_[1], hp_[2], hp_[3] };
                    if (HL8) {
#pragma unroll
                        for (int t = 0; t < 4; t++) if (t < cnt[p]) l[t] |= (rec[l[t]].x >> 26) << IXB;
                    }
#pragma unroll
                    for (int t = 0; t < 4; t++) if (t < cnt[p]) {
                        if ((l[t] ^ me) >> IXB) multi[p] = true;                   // another plane on the same pixel
                        if ((l[t] & IXM) < (me & IXM)) first[p] = false;           // compact index == event order
                    }
                    // two planes racing for an unlabelled pixel: ranked replay by the group's first event (below);
                    // on a labelled pixel the passive events of a third plane could matter: serial replay of the round
                    if (multi[p]) cx |= FS_LABEL(es[p]) < 0 ? 1 : 2;
                }
            }
        }
        if (cx) atomicOr(&s_cx[par], cx);
        lds_barrier();
        const int cxr = s_cx[par];
        const bool complex_round = (cxr & 2) != 0;
#pragma unroll
        for (int p = 0; p < FLOOD_NP; p++) if (p < npass && ev[p]) { hkeys[hs[p]] = -1; hcnt[hs[p]] = 0; }   // leave the hash empty for the next round
        FT(2)
        int total = 0;
        if (!complex_round) {
            // ---- closed form of a one-plane group, applied by its first event (AHCPlaneFitter.hpp:445-470) ----
#pragma unroll
            for (int p = 0; p < FLOOD_NP; p++) {
                if (p < npass && first[p] && !multi[p]) {
                    const int ep = (int)(eq[p] >> 26), trail = FS_LABEL(es[p]);
                    const bool closer = eok[p] && ecd[p] < ed[p];
                    if (eok[p] && trail >= 0 && ((simok[ep] >> trail) & 1ull)) { atomicOr(&adj[trail], 1ull << ep); atomicOr(&adj[ep], 1ull << trail); }
                    const int nl = closer ? ep : (trail < 0 ? max(trail - cnt[p], -6) : trail);
                    push[p] = closer;
                    if (nl != trail) state[eix[p]] = (es[p] & ~0xFFu) | ((unsigned)nl & 0xFFu);          // (closer: nl = ep != trail)
                }
            }
            if (cxr & 1) {
                n_ranked++;
                // ---- groups with two planes on an unlabelled pixel: every event publishes its distance, the
                //      group's first event replays the group in event order and hands the push flags back ----
#ifdef HVO_PEAC_TIMING
                ft[6]++;
#endif
#pragma unroll
                for (int p = 0; p < FLOOD_NP; p++) if (p < npass && multi[p]) hkeys[p * FLOOD_T + tid] = (int)(eok[p] ? __float_as_uint(ecd[p]) : 0xFFFFFFFFu);
                lds_barrier();
#pragma unroll
                for (int p = 0; p < FLOOD_NP; p++) {
                    if (p < npass && first[p] && multi[p]) {
                        const hl_t *hp_ = &hlist[hs[p] * FLOOD_HL];
                        unsigned l[4] = { hp_[0], hp_[1], hp_[2], hp_[3] };
                        if (HL8) {
#pragma unroll
                            for (int t = 0; t < 4; t++) if (t < cnt[p]) l[t] |= (rec[l[t]].x >> 26) << IXB;
                        }
                        int trail = FS_LABEL(es[p]); float dist = ed[p];
                        int last = -1;
                        for (int it = 0; it < cnt[p]; it++) {
                            unsigned best = 0; int c = 1 << 20;                     // next event of the group in event order
#pragma unroll
                            for (int t = 0; t < 4; t++) { const int ct = (int)(l[t] & IXM); if (t < cnt[p] && ct > last && ct < c) { c = ct; best = l[t]; } }
                            const int ep = (int)(best >> IXB);
                            last = c;
                            const unsigned okcd = (unsigned)hkeys[c];
                            const bool ok = okcd != 0xFFFFFFFFu; const float cd = __uint_as_float(okcd);
                            const bool live = !(trail <= -6 || trail == ep);
                            const bool closer = live && ok && cd < dist;
                            if (live && ok && trail >= 0 && ((simok[ep] >> trail) & 1ull)) { atomicOr(&adj[trail], 1ull << ep); atomicOr(&adj[ep], 1ull << trail); }
                            trail = closer ? ep : ((live && trail < 0) ? trail - 1 : trail);
                            dist = closer ? cd : dist;
                            hcnt[c] = closer ? 1 : 0;
                        }
                        state[eix[p]] = (es[p] & ~0xFFu) | ((unsigned)trail & 0xFFu);
                    }
                }
                lds_barrier();
#pragma unroll
                for (int p = 0; p < FLOOD_NP; p++) if (p < npass && multi[p]) { push[p] = hcnt[p * FLOOD_T + tid] != 0; hkeys[p * FLOOD_T + tid] = -1; hcnt[p * FLOOD_T + tid] = 0; }     // (the table is empty again)
            }
            // ---- ordered append: exclusive scan of the pushes in compact (= event) order ----
            unsigned long long bm[FLOOD_NP]; int off[FLOOD_NP];
#pragma unroll
            for (int p = 0; p < FLOOD_NP; p++) { bm[p] = 0; if (p < npass) bm[p] = __ballot(push[p]); }
            if (NW > 1) {
                if (lane == 0) for (int p = 0; p < FLOOD_NP; p++) psum[p * NW + wv] = __popcll(bm[p]);
                lds_barrier();
#pragma unroll
                for (int p = 0; p < FLOOD_NP; p++) {
                    off[p] = total;
                    for (int i = 0; i < NW; i++) { const int v = psum[p * NW + i]; if (i < wv) off[p] += v; total += v; }
                }
            } else {
#pragma unroll
                for (int p = 0; p < FLOOD_NP; p++) { off[p] = total; total += __popcll(bm[p]); }
            }
#pragma unroll
            for (int p = 0; p < FLOOD_NP; p++) if (p < npass && push[p]) {
                const int pos = nq + off[p] + __popcll(bm[p] & ltm);
                if (pos < r.qcap) queue[pos] = (int)eq[p];
            }
        } else {
            // ---- complex round: replay its entries in queue order on one thread (the reference loop) ----
            n_serial++;
#ifdef HVO_PEAC_TIMING
            ft[5]++;
#endif
            if (tid == 0) {
                int nqs = nq;
                for (int e = 0; e < nent; e++) {
                    const int qq = queue[kq + e];
                    const int ep = (int)((unsigned)qq >> 26), x0 = qq & 8191, y0 = (qq >> 13) & 8191;
                    for (int j = 0; j < 4; j++) {
                        const int x = x0 + (j == 0 ? -1 : j == 1 ? 1 : 0), y = y0 + (j == 2 ? -1 : j == 3 ? 1 : 0);
                        if (x < 0 || x >= w || y < 0 || y >= h) continue;
                        const int ix = FS_IDX(x, y, stw);
                        const uint32_t s = state[ix];
                        int trail = FS_LABEL(s); float dist;
                        if ((s & FS_VALID) || trail <= -6 || trail == ep) continue;
                        float cd; bool ok;
                        geom(pl[ep], x, y, (int)(s >> 16), trail, cd, ok, dist);
                        if (ok) {
                            if (trail >= 0 && ((simok[ep] >> trail) & 1ull)) { atomicOr(&adj[trail], 1ull << ep); atomicOr(&adj[ep], 1ull << trail); }
                            if (cd < dist) { trail = ep; dist = cd; if (nqs < r.qcap) queue[nqs] = FQ_PACK(x, y, ep); nqs++; }
                            else if (trail < 0) trail -= 1;
                        } else if (trail < 0) trail -= 1;
                        state[ix] = (s & ~0xFFu) | ((unsigned)trail & 0xFFu);
                    }
                }
                s_nq = nqs - nq;
            }
            lds_barrier();
            total = s_nq;
        }
        nq += total;
        if (nq > r.qcap) { nq = r.qcap; flags |= 32; }
        kq += nent;
        par ^= 1;
        __syncthreads();                                 // drains this round's state / queue stores
        FT(3)
    }
#ifdef HVO_PEAC_TIMING
    if (tid == 0) { for (int q = 0; q < 6; q++) atomicAdd(&g_peac_t[16 + q], ft[q]); for (int q = 6; q < 12; q++) atomicAdd(&g_peac_t[18 + q], ft[q]); atomicAdd(&g_peac_t[22], (unsigned long long)nq); atomicAdd(&g_peac_t[23], 1ull); }
#endif
    if (tid < MAX_PLANES) adj_out[(size_t)frame * MAX_PLANES + tid] = adj[tid];
    if (tid == 0) { meta[5] = nq; meta[7] = flags; meta[8] = n_rounds; meta[9] = n_ranked; meta[10] = n_serial; }
}

// k_peac_final: one last merge round over the still-valid coarse planes, plidmap, plane records
// (AHCPlaneFitter.hpp:321-340).  One wave per frame.
__global__ __launch_bounds__(64) void k_peac_final(RfArgs r, const unsigned long long *__restrict__ adj_in)
{
    __shared__ double hkey[MAX_PLANES]; __shared__ int hid[MAX_PLANES];
    __shared__ int lA[LCAP], lB[LCAP]; __shared__ double cm[64]; __shared__ int cN[64];
    const ClArgs &a = r.c;
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int nblk = a.nblk;
    int *meta = a.meta + (size_t)frame * 16;
    const int nold = meta[2];
    int *ext = a.extracted + (size_t)frame * 2 * MAX_PLANES;
    double *segD = a.segD + (size_t)frame * a.segcap * SEG_D;
    int *segI = a.segI + (size_t)frame * a.segcap * SEG_I;
    int *pool = a.pool + (size_t)frame * a.poolcap, *pool2 = a.pool2 + (size_t)frame * a.poolcap;
    if (meta[6]) { int *t = pool; pool = pool2; pool2 = t; }
    const int *isvalid = r.isvalid + (size_t)frame * MAX_PLANES;
    int *plidmap = r.plidmap + (size_t)frame * MAX_PLANES;
    const unsigned long long *adj = adj_in + (size_t)frame * MAX_PLANES;
    int flags = meta[3] | meta[7];
    Heap H; H.key = hkey; H.id = hid; H.n = 0;
    int nseg = meta[0], pooltop = meta[1];
    if (pooltop + nold * MAX_PLANES > a.poolcap - 2 * a.nblk) pool_gc(segI, nseg, pool, pool2, pooltop);
    if (lane == 0) {
        // fresh adjacency lists from the connect() calls, ascending seg id
        for (int p = 0; p < nold; p++) {
            int *si = segI + (size_t)ext[p] * SEG_I;
            si[3] = pooltop + p * MAX_PLANES; si[4] = 0; si[5] = MAX_PLANES;
        }
        for (int p = 0; p < nold; p++) {
            int ids[MAX_PLANES], n = 0;
            for (int q = 0; q < nold; q++) if ((adj[p] >> q) & 1ull) ids[n++] = ext[q];
            for (int x = 1; x < n; x++) { int v = ids[x], y = x - 1; while (y >= 0 && ids[y] > v) { ids[y + 1] = ids[y]; y--; } ids[y + 1] = v; }
            int *si = segI + (size_t)ext[p] * SEG_I;
            for (int x = 0; x < n; x++) pool[si[3] + x] = ids[x];
            si[4] = n;
        }
    }
    __syncthreads();
    for (int p = 0; p < nold; p++) if (isvalid[p]) { heap_push(H, segD[(size_t)ext[p] * SEG_D + 15], ext[p]); __syncthreads(); }
    pooltop += nold * MAX_PLANES;
    __syncthreads();
    int *fin = ext + MAX_PLANES;
    int nfin = 0;
    if (pooltop <= a.poolcap) ah_cluster_wave(a, frame, H, nseg, pooltop, pool, pool2, fin, nfin, lA, lB, cm, cN, flags);
    else flags |= 8;
    const int *parent = a.parent + (size_t)frame * nblk;
    if (lane == 0) {
        for (int p = 0; p < MAX_PLANES; p++) plidmap[p] = -1;
        for (int p = 0; p < nold; p++) {
            if (!isvalid[p]) continue;
            const int np_rid = ds_find_ro(parent, segI[(size_t)ext[p] * SEG_I + 1]);
            for (int j = 0; j < nfin; j++) if (segI[(size_t)fin[j] * SEG_I + 1] == np_rid) { plidmap[p] = j; break; }
        }
        hvo_plane *out = r.planes + (size_t)frame * MAX_PLANES;
        for (int j = 0; j < nfin; j++) {
            const double *sd = segD + (size_t)fin[j] * SEG_D;
            for (int k = 0; k < 3; k++) { out[j].normal[k] = sd[12 + k]; out[j].center[k] = sd[9 + k]; }
            out[j].mse = sd[15]; out[j].n_points = segI[(size_t)fin[j] * SEG_I]; out[j].rid = segI[(size_t)fin[j] * SEG_I + 1];
        }
        meta[3] = flags; meta[4] = nfin;
    }
}

// labels are written as int8 (plane ids < MAX_PLANES = 64, -1 = none); a thread takes one row of a state tile (32 contiguous
// bytes = eight pixels) and stores its labels as words when the width allows; the label slab of a frame is padded to a multiple of 4 bytes
__global__ __launch_bounds__(256) void k_peac_relabel(const uint32_t *__restrict__ state, int8_t *__restrict__ labels, const int *__restrict__ plidmap, int w, int h, size_t lframe)
{
    const int frame = blockIdx.y;
    const int tw = FS_TW(w), nrow = tw * FS_TH(h) * 4;
    const uint32_t *S = state + (size_t)frame * FS_FRAME(w, h);
    int8_t *L = labels + (size_t)frame * lframe;
    const int *pm = plidmap + (size_t)frame * MAX_PLANES;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nrow; i += gridDim.x * 256) {
        const int t = i >> 2, ty = t / tw, tx = t - ty * tw;
        const int x0 = tx * 8, y = ty * 4 + (i & 3);
        if (y >= h) continue;
        const uint4 a = reinterpret_cast<const uint4 *>(S + (size_t)i * 8)[0], b = reinterpret_cast<const uint4 *>(S + (size_t)i * 8)[1];
        const unsigned sx[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
        int lab[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int v = FS_LABEL(sx[k]); lab[k] = (v >= 0 && pm[v] >= 0) ? pm[v] : -1; }
        int8_t *dst = L + (size_t)y * w + x0;
        if ((w & 3) == 0) {                                         // (x0 and y * w are multiples of 4: whole words, each inside the row or outside it)
#pragma unroll
            for (int q = 0; q < 2; q++) if (x0 + 4 * q < w)
                reinterpret_cast<unsigned *>(dst)[q] = ((unsigned)lab[4 * q] & 0xFFu) | (((unsigned)lab[4 * q + 1] & 0xFFu) << 8) | (((unsigned)lab[4 * q + 2] & 0xFFu) << 16) | (((unsigned)lab[4 * q + 3] & 0xFFu) << 24);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) if (x0 + k < w) dst[k] = (int8_t)lab[k];
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
void peac_free(hvo_ctx *ctx)
{
    PeacPlan *P = plan_of(ctx);
    if (!P) return;
    void *ptrs[] = { P->d_depth, P->d_segD, P->d_segI, P->d_pool, P->d_pool2, P->d_parent, P->d_dsize, P->d_eflag, P->d_meta, P->d_extracted,
                     P->d_blkmap, P->d_labels, P->d_state, P->d_queue, P->d_plidmap, P->d_isvalid, P->d_planes, P->d_adj, P->d_hkey, P->d_m1k, P->d_hid, P->d_hot, P->d_lq };
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete P;
    ctx->peac = nullptr;
}

static int peac_build_plan(hvo_ctx *ctx, int w, int h, int batch);
// as orb_ensure_plan: a plan that fails half-way is freed, so its (w, h, batch) key never outlives its slabs
static int peac_ensure_plan(hvo_ctx *ctx, int w, int h, int batch)
{
    PeacPlan *P = plan_of(ctx);
    if (P && P->w == w && P->h == h && P->batch >= batch) return HVO_OK;
    if (w < 2 * WIN || h < 2 * WIN || w > 4096 || h > 4096) return HVO_ERR_UNSUPPORTED;
    peac_free(ctx);
    const int rc = peac_build_plan(ctx, w, h, batch);
    if (rc) peac_free(ctx);
    return rc;
}

static int peac_build_plan(hvo_ctx *ctx, int w, int h, int batch)
{
    PeacPlan *P = new PeacPlan();
    ctx->peac = P;
    P->kn.edges.read("HVO_PEAC_EDGES"); P->kn.gl.read("HVO_PEAC_GL"); P->kn.perm.read("HVO_PEAC_PERM"); P->kn.lend.read("HVO_PEAC_LEND"); P->kn.slots.read("HVO_PEAC_SLOTS"); P->kn.heads_maxn.read("HVO_PEAC_HEADS_MAXN");
    P->kn.heads.read("HVO_PEAC_HEADS"); P->kn.heads_big.read("HVO_PEAC_HEADS_BIG"); P->kn.poolcap.read("HVO_PEAC_POOLCAP"); P->kn.ldsq.read("HVO_PEAC_LDSQ");
    P->kn.flood_t.read("HVO_FLOOD_T"); P->kn.flood_epl.read("HVO_FLOOD_EPL"); P->kn.flood_perm.read("HVO_FLOOD_PERM");
    P->w = w; P->h = h; P->pitch = (w + 31) & ~31; P->Nw = w / WIN; P->Nh = h / WIN; P->nblk = P->Nw * P->Nh;
    P->segcap = 2 * P->nblk + 2 * MAX_PLANES; P->poolcap = 16 * P->nblk + 2 * MAX_PLANES * MAX_PLANES; P->qcap = 2 * w * h + 65536; P->batch = batch;
#define HVO_DEG2RAD(d) ((d) * 3.14159265358979323846 / 180.0)      /* MACRO_DEG2RAD, AHCParamSet.hpp:33: (d)*M_PI/180.0, in that order */
    // ParamSet: T_ang(P_INIT) clipped at z_near, similarityTh_merge, similarityTh_refine (AHCParamSet.hpp:68-76,113-134)
    const double factor = (HVO_DEG2RAD(90.0) - HVO_DEG2RAD(15.0)) / (4000.0 - 500.0);
    P->c15 = cos(factor * 500.0 + HVO_DEG2RAD(15.0) - factor * 500.0);
    P->ang_factor = factor; P->ang_near = HVO_DEG2RAD(15.0);
    P->c60 = cos(HVO_DEG2RAD(60.0)); P->c30 = cos(HVO_DEG2RAD(30.0));
#undef HVO_DEG2RAD
    const size_t B = batch, npix = (size_t)w * h;
#define PA(ptr, n) HVO_HIP(hipMalloc((void **)&(ptr), (n)))
    PA(P->d_depth, B * P->pitch * (h + 1) * sizeof(uint16_t));
    PA(P->d_segD, B * P->segcap * SEG_D * sizeof(double));
    PA(P->d_segI, B * P->segcap * SEG_I * sizeof(int));
    PA(P->d_pool, B * P->poolcap * sizeof(int) + 64);          // + 64: the list searches fetch eight entries at a time (peac_lend.inc)
    PA(P->d_pool2, B * P->poolcap * sizeof(int) + 64);
    PA(P->d_parent, B * P->nblk * sizeof(int)); PA(P->d_dsize, B * P->nblk * sizeof(int)); PA(P->d_eflag, B * P->nblk * sizeof(int));
    PA(P->d_meta, B * 16 * sizeof(int)); PA(P->d_extracted, B * 2 * MAX_PLANES * sizeof(int));
    PA(P->d_blkmap, B * P->nblk * sizeof(int)); PA(P->d_labels, B * ((npix + 3) & ~(size_t)3)); PA(P->d_state, B * FS_FRAME(w, h) * sizeof(uint32_t));
    PA(P->d_queue, B * P->qcap * sizeof(int));
    PA(P->d_plidmap, B * MAX_PLANES * sizeof(int)); PA(P->d_isvalid, B * MAX_PLANES * sizeof(int));
    PA(P->d_planes, B * MAX_PLANES * sizeof(hvo_plane));
    PA(P->d_adj, B * MAX_PLANES * sizeof(unsigned long long));
    PA(P->d_hot, B * P->segcap * 128);
    { const size_t n0 = (P->segcap + 255) / 256; PA(P->d_hkey, B * n0 * 256 * sizeof(double)); PA(P->d_m1k, B * n0 * 16 * sizeof(double)); PA(P->d_hid, B * n0 * 16 * sizeof(int)); PA(P->d_lq, B * n0 * 256 * sizeof(int)); }   // TQueue
#undef PA
    // stream-ordered fill: a null-stream hipMemset is not ordered against the non-blocking ctx stream
    HVO_HIP(hipMemsetAsync(P->d_depth, 0, B * P->pitch * (h + 1) * sizeof(uint16_t), ctx->s_peac));
    HVO_HIP(hipMemsetAsync(P->d_meta, 0, B * 16 * sizeof(int), ctx->s_peac));
    HVO_HIP(hipDeviceSynchronize());
    return HVO_OK;
}

bool peac_plan_covers(const hvo_ctx *ctx, int w, int h, int batch)
{
    const PeacPlan *P = static_cast<const PeacPlan *>(ctx->peac);
    return P && P->w == w && P->h == h && P->batch >= batch;
}
int peac_prepare(hvo_ctx *ctx, int w, int h, int batch, PeacView *v)
{
    int rc = peac_ensure_plan(ctx, w, h, batch);
    if (rc) return rc;
    PeacPlan *P = plan_of(ctx);
    v->d_depth = P->d_depth; v->pitch = P->pitch; v->dframe = (size_t)P->pitch * (h + 1); v->d_labels8 = P->d_labels; v->d_planes = P->d_planes;
    v->d_meta = P->d_meta; v->npix = w * h; v->max_planes = MAX_PLANES; v->lstride = ((size_t)w * h + 3) & ~(size_t)3;
    return HVO_OK;
}

int peac_upload(hvo_ctx *ctx, int n, const hvo_frame_in *in, int w, int h, bool sync)
{
    const hipStream_t cs = ctx->stage_depth_dst ? ctx->s_stage_up : hvo_copy_stream(ctx, ctx->s_peac);
    int rc = peac_ensure_plan(ctx, w, h, std::max(n, ctx->p.max_batch));
    if (rc) return rc;
    PeacPlan *P = plan_of(ctx);
    const size_t dframe = (size_t)P->pitch * (h + 1);
    uint16_t *const dst = ctx->stage_depth_dst ? ctx->stage_depth_dst : P->d_depth;      // the resident depth slab, or the staging slab of a double-buffered batch
    for (int f = 0; f < n; f++) if (!in[f].depth) return HVO_ERR_INVALID_ARG;
    // dense host frames go up in runs of evenly spaced frames: one 2-D copy per run whose rows are whole frames (see orb_upload)
    const bool dense = P->pitch == w && !ctx->kn_upload_single.off();
    const int dstride = (int)(w * sizeof(uint16_t));
    for (int f = 0; f < n;) {
        int run = 1;
        if (dense && in[f].depth_stride == dstride && f + 1 < n && in[f + 1].depth_stride == dstride) {
            const ptrdiff_t step = (const char *)in[f + 1].depth - (const char *)in[f].depth;
            if (step >= (ptrdiff_t)((size_t)w * h * sizeof(uint16_t))) {
                while (f + run < n && in[f + run].depth_stride == dstride && (const char *)in[f + run].depth - (const char *)in[f + run - 1].depth == step) run++;
                if (run > 1) HVO_HIP(hipMemcpy2DAsync(dst + (size_t)f * dframe, dframe * sizeof(uint16_t), in[f].depth, (size_t)step, (size_t)w * h * sizeof(uint16_t), run, hipMemcpyHostToDevice, cs));
            }
        }
        if (run == 1)
            HVO_HIP(hipMemcpy2DAsync(dst + (size_t)f * dframe, P->pitch * sizeof(uint16_t), in[f].depth, in[f].depth_stride,
                                     (size_t)w * sizeof(uint16_t), h, hipMemcpyHostToDevice, cs));
        f += run;
    }
    if (sync) HVO_HIP(hipStreamSynchronize(cs));
    return HVO_OK;
}

int peac_run(hvo_ctx *ctx, int n)
{
    PeacPlan *P = plan_of(ctx);
    if (!P || n < 1 || n > P->batch) return HVO_ERR_INVALID_ARG;
    hipStream_t st = hvo_stream_peac(ctx);
    const size_t dframe = (size_t)P->pitch * (P->h + 1);
    const hvo_params &p = ctx->p;
    HVO_HIP(hipMemsetAsync(P->d_isvalid, 0, (size_t)n * MAX_PLANES * sizeof(int), st));
    int id = hvo_prof_begin(ctx, "peac_blocks", st);
    hipLaunchKernelGGL(k_peac_blocks, dim3((P->nblk + 63) / 64, n), dim3(64), 0, st, P->d_depth, dframe, P->pitch, P->w, P->h, P->Nw, P->nblk,
                       p.fx, p.fy, p.cx, p.cy, p.depth_map_factor, P->d_segD, P->d_segI, P->segcap, (HotNode *)P->d_hot);
    hvo_prof_end(ctx, id);
    ClArgs a;
    a.segD = P->d_segD; a.segI = P->d_segI; a.pool = P->d_pool; a.pool2 = P->d_pool2; a.parent = P->d_parent; a.dsize = P->d_dsize; a.eflag = P->d_eflag;
    a.meta = P->d_meta; a.extracted = P->d_extracted; a.segcap = P->segcap; a.poolcap = P->poolcap; a.nblk = P->nblk; a.Nw = P->Nw; a.Nh = P->Nh;
    a.c15 = P->c15; a.c60 = P->c60; a.hot = (HotNode *)P->d_hot; a.perm = nullptr; a.tqK = P->d_hkey; a.tqM1k = P->d_m1k; a.tqM1i = P->d_hid; a.tqLq = P->d_lq; a.tq_n0 = (P->segcap + 255) / 256;
    a.ang_factor = P->ang_factor; a.ang_near = P->ang_near;
    if (ctx->sched == 6 && !ctx->serialize) {               // experiment: the streaming kernels of the other stages first, then the serial ones together
        if (ctx->fast_recorded) HVO_HIP(hipStreamWaitEvent(st, ctx->ev_fast, 0));
        if (ctx->lsd_pre_recorded) HVO_HIP(hipStreamWaitEvent(st, ctx->ev_lsd_pre, 0));
    }
    id = hvo_prof_begin(ctx, "peac_cluster", st);
    a.edges_done = 0;
    {
        const size_t elds = (size_t)P->nblk * 34 + 16;
        // HVO_PEAC_EDGES=0: the passes stay inside the clustering kernels (A/B runs, tests)
        if (elds <= 150 * 1024 && !P->kn.edges.off()) {
            if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_peac_edges), elds)) return HVO_ERR_HIP;
            hipLaunchKernelGGL(k_peac_edges, dim3(n), dim3(512), elds, st, a, n);
            a.edges_done = 1;
        }
    }
    {
        // 4 frames per wave pay off once the wave slots are saturated (measured: >= ~3000 resident frames);
        // HVO_PEAC_GL forces a group width (tests run the 16-lane path on small batches with it)
        const int gl = P->kn.gl.or_(-1);
        const int use = gl > 0 ? gl : (n >= 3072 ? 16 : 64);
        const size_t lq = (size_t)a.tq_n0 * 12;                  // LDS per group: the queue's top level
        // the waves that share a SIMD are a fixed stride apart: their frames are decorrelated (hvo_frame_perm over the waves; the
        // frames of one wave stay consecutive, lockstep likes them alike); HVO_PEAC_PERM=0 for A/B runs
        a.perm = hvo_frame_perm(ctx, (n + (64 / use) - 1) / (64 / use));
        if (P->kn.perm.off()) a.perm = nullptr;
        a.tq_lds_keys = 0;
        // a handful of frames (the latency case): several queue heads per round, one wave each (peac_heads.inc); HVO_PEAC_HEADS = 0 / 2 / 3 / 4
        const int heads_max = P->kn.heads_maxn.or_(256);
        // three heads + the queue wave = one wave per SIMD of the frame's CU: 256 frames fill the chip exactly, and a lone frame loses nothing
        // against four (a round costs 7.3 instead of 8.0 us for 2.12 instead of 2.34 pops; batch256 9.6 against 9.0 k frames/s)
        // frames whose per-id keys and list headers do not fit LDS (1280x960: 24 704 node ids) take the BIG form: bucket minima, front and
        // conflict bitmaps in LDS, keys in global memory, headers in the node records
        const bool heads_fit = a.tq_n0 * 16 <= 64 * MH_MAXE, heads_big = !heads_fit && a.tq_n0 * 16 <= 64 * MH_MAXE_BIG;
        int heads = (gl <= 0 && n <= heads_max && (heads_fit || heads_big)) ? 3 : 0;
        if (P->kn.heads.set && (heads_fit || heads_big)) heads = P->kn.heads.v;
        // The LDS form takes 108 KB: one workgroup per CU.  From ~130 frames on the workgroups need (nearly) every CU AT ONCE, and whichever
        // streaming kernel holds more than 52 KB of a CU's LDS at that moment (the ORB tiles, the LSD preamble, the flood) sends one of them
        // into a second turn: batch256 took 27.5 or 33 ms from step to step (profiles/r04_batch256_modes.txt).  The BIG form (32 KB) shares
        // a CU: 10 % slower for a lone frame, no second mode at 256 frames.
        bool big = heads_big || n > 128;
        if (P->kn.heads_big.set && heads_fit) big = P->kn.heads_big.v != 0;      // tests: the BIG form on a frame that would fit
        if (heads >= 2 && heads <= 4) {
            ClArgs b = a;
            if (P->kn.poolcap.set && P->kn.poolcap.v >= 7 * a.nblk && P->kn.poolcap.v < a.poolcap) b.poolcap = P->kn.poolcap.v;   // tests: force the pool's compaction
            const size_t segpad = (size_t)a.tq_n0 * 256;
            const size_t lds = (big ? 0 : segpad * 8) + (size_t)a.tq_n0 * 16 * 12 + segpad / 8 * 4 + 4 * sizeof(MhHead) + 32 + (big ? 0 : segpad * 8) + 512;
            {
                const void *kf = big ? (heads == 2 ? reinterpret_cast<const void *>(k_peac_cluster_heads<2, true>) : heads == 3 ? reinterpret_cast<const void *>(k_peac_cluster_heads<3, true>) : reinterpret_cast<const void *>(k_peac_cluster_heads<4, true>))
                                     : (heads == 2 ? reinterpret_cast<const void *>(k_peac_cluster_heads<2, false>) : heads == 3 ? reinterpret_cast<const void *>(k_peac_cluster_heads<3, false>) : reinterpret_cast<const void *>(k_peac_cluster_heads<4, false>));
                if (hvo_ensure_dyn_lds(kf, lds)) return HVO_ERR_HIP;
            }
            if (big) {
                if (heads == 2) hipLaunchKernelGGL((k_peac_cluster_heads<2, true>), dim3(n), dim3(192), lds, st, b, n);
                else if (heads == 3) hipLaunchKernelGGL((k_peac_cluster_heads<3, true>), dim3(n), dim3(256), lds, st, b, n);
                else hipLaunchKernelGGL((k_peac_cluster_heads<4, true>), dim3(n), dim3(320), lds, st, b, n);
            } else {
                if (heads == 2) hipLaunchKernelGGL((k_peac_cluster_heads<2, false>), dim3(n), dim3(192), lds, st, b, n);
                else if (heads == 3) hipLaunchKernelGGL((k_peac_cluster_heads<3, false>), dim3(n), dim3(256), lds, st, b, n);
                else hipLaunchKernelGGL((k_peac_cluster_heads<4, false>), dim3(n), dim3(320), lds, st, b, n);
            }
        }
        else if (use == 64) {
            size_t lds = lq;
            bool ldsq = n <= 256;                                  // at most one such frame per CU
            if (P->kn.ldsq.set) ldsq = P->kn.ldsq.v != 0;
            const size_t full = ((lq + 15) & ~(size_t)15) + (size_t)a.tq_n0 * (256 * 8 + 16 * 12);
            if (ldsq && full <= 150 * 1024) {
                a.tq_lds_keys = 1; lds = full;
                if (hvo_ensure_dyn_lds(reinterpret_cast<const void *>(k_peac_cluster<64, false>), lds)) return HVO_ERR_HIP;
            }
            hipLaunchKernelGGL((k_peac_cluster<64, false>), dim3(n), dim3(64), lds, st, a, n);
        }
        else if (use == 32) hipLaunchKernelGGL((k_peac_cluster<32, false>), dim3((n + 1) / 2), dim3(64), 2 * lq, st, a, n);
        else if (use == 16 && P->kn.poolcap.set && P->kn.poolcap.v >= 7 * a.nblk && P->kn.poolcap.v < a.poolcap) {   // tests: force the pool's compaction in the four-frames-per-wave kernels
            ClArgs b = a; b.poolcap = P->kn.poolcap.v;
            if (!P->kn.slots.off() && !P->kn.lend.off() && a.nblk < 32768 && P->d_lq) hipLaunchKernelGGL(k_peac_cluster_slots, dim3((n + 3) / 4), dim3(64), 4 * lq, st, b, n);
            else if (P->kn.lend.off()) hipLaunchKernelGGL((k_peac_cluster<16, false>), dim3((n + 3) / 4), dim3(64), 4 * lq, st, b, n);
            else hipLaunchKernelGGL((k_peac_cluster<16, true>), dim3((n + 3) / 4), dim3(64), 4 * lq, st, b, n);
        }
        else if (!P->kn.slots.off() && !P->kn.lend.off() && a.nblk < 32768 && P->d_lq) hipLaunchKernelGGL(k_peac_cluster_slots, dim3((n + 3) / 4), dim3(64), 4 * lq, st, a, n);   // HVO_PEAC_SLOTS=0: a new record per merge
        else if (P->kn.lend.off()) hipLaunchKernelGGL((k_peac_cluster<16, false>), dim3((n + 3) / 4), dim3(64), 4 * lq, st, a, n);
        else hipLaunchKernelGGL((k_peac_cluster<16, true>), dim3((n + 3) / 4), dim3(64), 4 * lq, st, a, n);   // HVO_PEAC_LEND=0: the lanes of a frame stay with it
    }
    hvo_prof_end(ctx, id);
    id = hvo_prof_begin(ctx, "peac_refine", st);
    hipLaunchKernelGGL(k_peac_blkmap, dim3(n, n <= 64 ? 32 : n <= 1024 ? 4 : 1), dim3(256), 0, st, P->d_parent, P->d_dsize, P->d_segI, P->d_extracted, P->d_meta, P->d_blkmap,
                       P->d_isvalid, P->d_state, P->d_depth, dframe, P->pitch, P->nblk, P->Nw, P->Nh, P->w, P->h, P->segcap);
    RfArgs r;
    r.c = a; r.depth = P->d_depth; r.dframe = dframe; r.pitch = P->pitch; r.w = P->w; r.h = P->h;
    r.fx = p.fx; r.fy = p.fy; r.cx = p.cx; r.cy = p.cy; r.dfac = p.depth_map_factor;
    r.blkmap = P->d_blkmap; r.isvalid = P->d_isvalid; r.state = P->d_state; r.queue = P->d_queue; r.perm = nullptr;
    r.qcap = P->qcap; r.plidmap = P->d_plidmap; r.planes = P->d_planes; r.c30 = P->c30;
    {
        // threads per frame (one queue entry = 4 events per thread and round); HVO_FLOOD_T overrides
        const int flood_t = P->kn.flood_t.or_(-1);
        // measured: 256 threads per frame up to ~4096 resident frames, one wave per frame (less LDS, all frames in flight) beyond
        // a lone frame (the latency case) takes 512: the flood's rounds are serial, so its time goes with the events a round retires
        const int ft = flood_t > 0 ? flood_t : (n >= 6144 ? 64 : n <= 8 ? 512 : 256);
        const int fe = P->kn.flood_epl.or_(1);             // HVO_FLOOD_EPL: queue entries per thread and round (one-wave variant only)
        r.perm = hvo_frame_perm(ctx, n);                   // one wave per frame for its whole life: frames of a SIMD decorrelated
        if ((ctx->sched == 5 || ctx->sched == 7) && ctx->fast_recorded && !ctx->serialize) HVO_HIP(hipStreamWaitEvent(st, ctx->ev_fast, 0));      // experiment: the flood takes all LDS, k_fast_cells needs some
        if (P->kn.flood_perm.off()) r.perm = nullptr;
        if (ft == 64 && fe == 2) hipLaunchKernelGGL((k_peac_flood<64, 2>), dim3(n), dim3(64), 0, st, r, P->d_adj);
        else if (ft == 64) hipLaunchKernelGGL((k_peac_flood<64, 1>), dim3(n), dim3(64), 0, st, r, P->d_adj);
        else if (ft == 512) hipLaunchKernelGGL((k_peac_flood<512, 1>), dim3(n), dim3(512), 0, st, r, P->d_adj);
        else if (ft == 256) hipLaunchKernelGGL((k_peac_flood<256, 1>), dim3(n), dim3(256), 0, st, r, P->d_adj);
        else hipLaunchKernelGGL((k_peac_flood<128, 1>), dim3(n), dim3(128), 0, st, r, P->d_adj);
    }
    hipLaunchKernelGGL(k_peac_final, dim3(n), dim3(64), 0, st, r, P->d_adj);
    hipLaunchKernelGGL(k_peac_relabel, dim3(64, n), dim3(256), 0, st, P->d_state, P->d_labels, P->d_plidmap, P->w, P->h, ((size_t)P->w * P->h + 3) & ~(size_t)3);
    hvo_prof_end(ctx, id);
    HVO_HIP(hipGetLastError());
    return HVO_OK;
}

int peac_download(hvo_ctx *ctx, int n, hvo_frame_out *out)
{
    const hipStream_t cs = hvo_copy_stream(ctx, ctx->s_peac);
    PeacPlan *P = plan_of(ctx);
    if (!P) return HVO_ERR_INVALID_ARG;
    std::vector<int> meta((size_t)n * 16);
    HVO_HIP(hipMemcpyAsync(meta.data(), P->d_meta, meta.size() * sizeof(int), hipMemcpyDeviceToHost, cs));
    HVO_HIP(hipStreamSynchronize(cs));
    const size_t npix = (size_t)P->w * P->h;
    std::vector<void *> dl(n, nullptr), dl8(n, nullptr), dp(n, nullptr); std::vector<size_t> bl(n, 0), bl8(n, 0), bp(n, 0);
    bool any32 = false, any8 = false;
    for (int f = 0; f < n; f++) { any32 |= out[f].labels != nullptr; any8 |= out[f].labels8 != nullptr; }
    for (int f = 0; f < n; f++) {
        const int nfin = meta[(size_t)f * 16 + 4], flags = meta[(size_t)f * 16 + 3];
        if (flags) out[f].status = HVO_ERR_CAPACITY;
        if (out[f].labels) { dl[f] = out[f].labels; bl[f] = npix; }          // int8 on the wire, widened to int32 while scattering
        if (out[f].labels8) { dl8[f] = out[f].labels8; bl8[f] = npix; }
        int m = nfin;
        if (out[f].planes) {
            if (m > out[f].pl_cap) { m = out[f].pl_cap; out[f].status = HVO_ERR_CAPACITY; }
            if (m > 0) { dp[f] = out[f].planes; bp[f] = (size_t)m * sizeof(hvo_plane); }
        }
        out[f].n_planes = m;
    }
    int rc = HVO_OK;
    if (any32 && (rc = hvo_staged_d2h(ctx, cs, P->d_labels, (npix + 3) & ~(size_t)3, n, dl.data(), bl.data(), 1))) return rc;
    if (any8 && (rc = hvo_staged_d2h(ctx, cs, P->d_labels, (npix + 3) & ~(size_t)3, n, dl8.data(), bl8.data(), 0))) return rc;
    if ((rc = hvo_staged_d2h(ctx, cs, P->d_planes, (size_t)MAX_PLANES * sizeof(hvo_plane), n, dp.data(), bp.data()))) return rc;
    HVO_HIP(hipStreamSynchronize(cs));
    return HVO_OK;
}

extern "C" int hvo_compute_planes(hvo_ctx *ctx, const uint16_t *depth, int w, int h, int stride,
                                  int32_t *labels, hvo_plane *planes, int cap, int *n)
{
    if (!ctx || !n) return HVO_ERR_INVALID_ARG;
    *n = 0;
    if (!depth || w <= 0 || h <= 0) return HVO_ERR_BAD_DTYPE;      // PlaneExtractor.cpp:34-38: empty image -> false
    if (cap < 0 || stride < w * 2) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    hvo_frame_in in; memset(&in, 0, sizeof(in));
    in.depth = depth; in.depth_stride = stride;
    int rc = peac_upload(ctx, 1, &in, w, h);
    if (rc) return rc;
    ctx->last_stages = 0;                                  // slot 0 of the resident batch has been overwritten
    for (int i = 0; i < ctx->nprof; i++) ctx->prof[i].used = false;
    if ((rc = peac_run(ctx, 1))) return rc;
    hvo_frame_out out; memset(&out, 0, sizeof(out));
    out.labels = labels; out.planes = planes; out.pl_cap = cap;
    if ((rc = peac_download(ctx, 1, &out))) return rc;
    *n = out.n_planes;
    return out.status;
}

// diagnostics: the frame's 16 bookkeeping words of the last plane run: [0] AHC segments, [2] coarse planes, [3] capacity flags,
// [4] final planes, [5] flood queue entries, [8] flood rounds, [9] rounds with ranked two-plane groups, [10] rounds replayed serially
extern "C" int hvo_debug_peac_stats(hvo_ctx *ctx, int frame, int *out16)
{
    PeacPlan *P = ctx ? plan_of(ctx) : nullptr;
    if (!P || !out16 || frame < 0 || frame >= P->batch) return HVO_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return HVO_ERR_NO_DEVICE;
    HVO_HIP(hipMemcpyAsync(out16, P->d_meta + (size_t)frame * 16, 16 * sizeof(int), hipMemcpyDeviceToHost, ctx->s_peac));
    HVO_HIP(hipStreamSynchronize(ctx->s_peac));
    return HVO_OK;
}

#ifdef HVO_PEAC_TIMING
// diagnostics build only (make DEFS=-DHVO_PEAC_TIMING): accumulated clock64() ticks per AHC phase
extern "C" int hvo_debug_peac_timing(unsigned long long *out32, int reset)
{
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_peac_t), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(g_peac_t), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

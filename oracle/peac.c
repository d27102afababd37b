/*
 * peac.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * CPU restatement of the reference's plane extraction:
 *   PlaneDetection::readDepthImage      src/PlaneExtractor.cpp:26-58, include/PlaneExtractor.h:25-33
 *   ahc::PlaneFitter::run               include/peac/AHCPlaneFitter.hpp:211-260
 *     initGraph                         include/peac/AHCPlaneFitter.hpp:786-972
 *     ahCluster                         include/peac/AHCPlaneFitter.hpp:983-1189
 *     refineDetails                     include/peac/AHCPlaneFitter.hpp:299-379
 *     findBlockMembership               include/peac/AHCPlaneFitter.hpp:485-587
 *     floodFill                         include/peac/AHCPlaneFitter.hpp:428-476
 *   ahc::PlaneSeg (+Stats)              include/peac/AHCPlaneSeg.hpp:52-409
 *   ahc::ParamSet                       include/peac/AHCParamSet.hpp:43-146   (defaults; NOT configurable)
 *   DisjointSet                         include/peac/DisjointSet.hpp:31-97
 *
 * Reference quirk kept on purpose: the cloud is in METRES (u16 * 1/5000) while the
 * ParamSet defaults are for millimetres, so T_mse never rejects and T_ang(P_INIT) clips to cos 15deg.
 *
 * Determinism rules where the reference depends on heap addresses / unspecified order:
 *   - std::set<PlaneSeg*> iteration (AHCPlaneSeg.hpp:188, AHCPlaneFitter.hpp:1031): ascending node
 *     creation order.
 *   - std::priority_queue ties on mse and std::sort ties on N: creation order.
 *   - LA::eig33sym (Eigen::SelfAdjointEigenSolver, eig33sym.hpp:70-74, Eigen not vendored): ASSUMED
 *     any backward-stable symmetric 3x3 solver.  Stats::compute only uses the smallest eigenpair, which
 *     orc_eig33_smallest computes (Laguerre iteration + adjugate column, fixed operation order; the HIP
 *     kernels run the identical sequence).  The cyclic Jacobi solver orc_eig33sym is kept as the
 *     high-accuracy cross-check of that routine.
 *   - membershipImg keeps negative "trail" counters for unlabelled pixels; the oracle reports them
 *     all as -1.
 */
#include "oracle.h"
#define _GNU_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ---- ParamSet defaults (AHCParamSet.hpp:68-76) ---- */
#define P_DEPTH_SIGMA 1.6e-6
#define P_STDTOL_INIT 5.0
#define P_STDTOL_MERGE 8.0
#define P_Z_NEAR 500.0
#define P_Z_FAR 4000.0
#define P_DEPTH_ALPHA 0.04
#define P_DEPTH_CHANGE_TOL 0.02
#define WIN 10                 /* windowWidth = windowHeight = 10 (AHCPlaneFitter.hpp:155) */
#define MIN_SUPPORT 3000       /* AHCPlaneFitter.hpp:154 */
#define MAX_STEP 100000

static double deg2rad(double d) { return d * M_PI / 180.0; }

static double T_mse_init(double z) { double t = P_DEPTH_SIGMA * z * z + P_STDTOL_INIT; return t * t; }
static double T_mse_merge(double z) { double t = P_DEPTH_SIGMA * z * z + P_STDTOL_MERGE; return t * t; }
static double T_ang_init(double z)
{
    double cz = z;
    if (cz < P_Z_NEAR) cz = P_Z_NEAR;
    if (cz > P_Z_FAR) cz = P_Z_FAR;
    const double factor = (deg2rad(90.0) - deg2rad(15.0)) / (P_Z_FAR - P_Z_NEAR);
    return cos(factor * cz + deg2rad(15.0) - factor * P_Z_NEAR);
}

/* symmetric 3x3 eigen-decomposition, eigenvalues ascending, V[:,i] <-> s[i].
 * Cyclic Jacobi, fixed order (0,1),(0,2),(1,2), relative convergence test, at most 30 sweeps. */
void orc_eig33sym(const double Kin[3][3], double s[3], double V[3][3])
{
    double a[3][3], v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = Kin[i][j];
    static const int PQ[3][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 } };
    for (int sweep = 0; sweep < 30; sweep++) {
        /* converged once the off-diagonal mass is below 1e-13 of the diagonal mass: the eigenvalues are then
         * exact to second order (~1e-26 relative) and the eigenvectors to ~1e-13, three sweeps in practice */
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off <= 1e-13 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;
        for (int r = 0; r < 3; r++) {
            const int p = PQ[r][0], q = PQ[r][1];
            const double apq = a[p][q];
            if (apq == 0.0) continue;
            /* t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)), theta = d / h, written with one division */
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
            for (int i = 0; i < 3; i++) {
                const double vip = v[i][p], viq = v[i][q];
                v[i][p] = c * vip - sn * viq;
                v[i][q] = sn * vip + c * viq;
            }
        }
    }
    int o[3] = { 0, 1, 2 };
    double d[3] = { a[0][0], a[1][1], a[2][2] };
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++)
        if (d[o[j]] < d[o[i]]) { int t = o[i]; o[i] = o[j]; o[j] = t; }
    for (int i = 0; i < 3; i++) { s[i] = d[o[i]]; for (int r = 0; r < 3; r++) V[r][i] = v[r][o[i]]; }
}

/* Smallest eigenpair of a symmetric positive semi-definite 3x3 matrix: all that Stats::compute uses of LA::eig33sym
 * (AHCPlaneSeg.hpp:139-153: normal = V[:,0] flipped towards the camera, mse = s[0] / N, curvature = s[0] / (s[0]+s[1]+s[2]),
 * and s[0]+s[1]+s[2] = trace(K)).
 * lambda0: Laguerre's iteration on q(l) = det(l I - K) from l = 0.  For a polynomial with real roots Laguerre's step from
 * below the smallest root never passes it, so the iterates rise monotonically to lambda0 (cubic convergence for a simple
 * root, two or three steps for plane-like covariances); q, q' and q'' come from K - l I itself:
 *     q = -det(K - l I),  q' = sum of its principal 2x2 minors,  q'' = -2 trace(K - l I),
 *     l <- l + 3 det / (q' + sqrt(4 q'^2 - 6 q q'')).
 * It stops when a step no longer raises l by more than one ulp of trace(K) (or after 8 steps).  Eigenvector: adj(K - lambda0 I)
 * has rank one (= c v v^T), so the column holding its largest diagonal minor, normalised, is v.
 * Accuracy: |lambda0 - exact| <= ~2e-16 trace(K), the backward-stable bound that the reference's own solver
 * (Eigen::SelfAdjointEigenSolver, tridiagonal QL; eig33sym.hpp:70-74) meets; measured against the cyclic Jacobi solver
 * orc_eig33sym over 2e5 random covariances of noisy planar patches: max 2.3e-16 trace(K), normals within 5e-8 rad
 * (tests/test_oracle_known_answers.py).  Only + - * / sqrt in a fixed order: the HIP kernels run the identical sequence. */
void orc_eig33_smallest(const double K[3][3], double *lambda0, double v[3])
{
    const double a = K[0][0], b = K[1][1], c = K[2][2], d = K[0][1], e = K[1][2], f = K[0][2];
    const double tol = 2.220446049250313e-16 * (a + b + c);
    double l = 0.0;
    for (int it = 0; it < 8; it++) {
        const double A = a - l, B = b - l, C = c - l;
        const double m1 = B * C - e * e, m2 = A * C - f * f, m3 = A * B - d * d;
        const double det = A * m1 - d * (d * C - e * f) + f * (d * e - B * f);
        const double dq = m1 + m2 + m3;
        if (!(dq > 0.0)) break;
        const double q2 = -2.0 * (A + B + C);
        double disc = 4.0 * dq * dq + 6.0 * det * q2;
        if (disc < 0.0) disc = 0.0;
        const double ln = l + 3.0 * det / (dq + sqrt(disc));
        if (!(ln > l)) break;
        const double step = ln - l;
        l = ln;
        if (step <= tol) break;
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
    else { v[0] = 0.0; v[1] = 0.0; v[2] = 1.0; }            /* K is a multiple of the identity: any direction */
    *lambda0 = l;
}

typedef struct { double sx, sy, sz, sxx, syy, szz, sxy, syz, sxz; int N; } stats_t;

typedef struct {
    stats_t st;
    int rid, N, nouse;
    double mse, center[3], normal[3], curvature;
    int *nbs; int nnb, capnb;     /* sorted ascending node ids */
} seg_t;

/* Stats::compute (AHCPlaneSeg.hpp:125-156) */
static void stats_compute(const stats_t *s, double center[3], double normal[3], double *mse, double *curv)
{
    const double sc = 1.0 / s->N;
    center[0] = s->sx * sc; center[1] = s->sy * sc; center[2] = s->sz * sc;
    double K[3][3] = {
        { s->sxx - s->sx * s->sx * sc, s->sxy - s->sx * s->sy * sc, s->sxz - s->sx * s->sz * sc },
        { 0, s->syy - s->sy * s->sy * sc, s->syz - s->sy * s->sz * sc },
        { 0, 0, s->szz - s->sz * s->sz * sc } };
    K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
    double l0, v[3];
    orc_eig33_smallest(K, &l0, v);
    if (v[0] * center[0] + v[1] * center[1] + v[2] * center[2] <= 0) {
        normal[0] = v[0]; normal[1] = v[1]; normal[2] = v[2];
    } else {
        normal[0] = -v[0]; normal[1] = -v[1]; normal[2] = -v[2];
    }
    *mse = l0 * sc;
    *curv = l0 / (K[0][0] + K[1][1] + K[2][2]);
}

typedef struct {
    int w, h, Nw, Nh;
    const double *X, *Y, *Z;        /* organised cloud (readDepthImage) */
    seg_t *seg; int nseg, capseg;
    int *parent, *dsize;            /* DisjointSet */
    /* heap of (mse, id) */
    int *heap; int nheap, capheap;
    int *extracted; int nextracted;
} fitter_t;

static int cloud_get(const fitter_t *f, int row, int col, double *x, double *y, double *z)
{
    const int i = row * f->w + col;
    *z = f->Z[i];
    if (*z == 0 || isnan(*z)) return 0;
    *x = f->X[i]; *y = f->Y[i];
    return 1;
}

static int ds_find(fitter_t *f, int x) { while (f->parent[x] != x) { f->parent[x] = f->parent[f->parent[x]]; x = f->parent[x]; } return x; }
/* NOTE: path halving instead of the reference's full recursion -- same roots/sizes, the parent
 * array is internal. */
static int ds_union(fitter_t *f, int x, int y)
{
    int xr = ds_find(f, x), yr = ds_find(f, y);
    if (xr == yr) return xr;
    if (f->dsize[xr] < f->dsize[yr]) { f->parent[xr] = yr; f->dsize[yr] += f->dsize[xr]; return yr; }
    f->parent[yr] = xr; f->dsize[xr] += f->dsize[yr]; return xr;
}

static int new_seg(fitter_t *f)
{
    if (f->nseg == f->capseg) { f->capseg = f->capseg * 2 + 1024; f->seg = (seg_t *)realloc(f->seg, sizeof(seg_t) * f->capseg); }
    memset(&f->seg[f->nseg], 0, sizeof(seg_t));
    return f->nseg++;
}

static void nb_insert(seg_t *s, int id)
{
    int lo = 0, hi = s->nnb;
    while (lo < hi) { int m = (lo + hi) / 2; if (s->nbs[m] < id) lo = m + 1; else hi = m; }
    if (lo < s->nnb && s->nbs[lo] == id) return;
    if (s->nnb == s->capnb) { s->capnb = s->capnb * 2 + 8; s->nbs = (int *)realloc(s->nbs, sizeof(int) * s->capnb); }
    memmove(s->nbs + lo + 1, s->nbs + lo, sizeof(int) * (s->nnb - lo));
    s->nbs[lo] = id; s->nnb++;
}
static void nb_erase(seg_t *s, int id)
{
    for (int i = 0; i < s->nnb; i++) if (s->nbs[i] == id) { memmove(s->nbs + i, s->nbs + i + 1, sizeof(int) * (s->nnb - i - 1)); s->nnb--; return; }
}
static void connect_seg(fitter_t *f, int a, int b) { nb_insert(&f->seg[a], b); nb_insert(&f->seg[b], a); }
static void disconnect_all(fitter_t *f, int a)
{
    seg_t *s = &f->seg[a];
    for (int i = 0; i < s->nnb; i++) nb_erase(&f->seg[s->nbs[i]], a);
    s->nnb = 0;
}
static double normal_similarity(const seg_t *a, const seg_t *b)
{
    return fabs(a->normal[0] * b->normal[0] + a->normal[1] * b->normal[1] + a->normal[2] * b->normal[2]);
}

/* min-heap on (mse, id) */
static int heap_less(const fitter_t *f, int a, int b)
{
    double ma = f->seg[a].mse, mb = f->seg[b].mse;
    return ma < mb || (ma == mb && a < b);
}
static void heap_push(fitter_t *f, int id)
{
    if (f->nheap == f->capheap) { f->capheap = f->capheap * 2 + 1024; f->heap = (int *)realloc(f->heap, sizeof(int) * f->capheap); }
    int i = f->nheap++;
    f->heap[i] = id;
    while (i > 0) { int p = (i - 1) / 2; if (heap_less(f, f->heap[i], f->heap[p])) { int t = f->heap[i]; f->heap[i] = f->heap[p]; f->heap[p] = t; i = p; } else break; }
}
static int heap_pop(fitter_t *f)
{
    int top = f->heap[0];
    f->heap[0] = f->heap[--f->nheap];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < f->nheap && heap_less(f, f->heap[l], f->heap[m])) m = l;
        if (r < f->nheap && heap_less(f, f->heap[r], f->heap[m])) m = r;
        if (m == i) break;
        int t = f->heap[i]; f->heap[i] = f->heap[m]; f->heap[m] = t; i = m;
    }
    return top;
}

/* PlaneSeg block constructor (AHCPlaneSeg.hpp:210-285), INIT_STRICT */
static int init_block(fitter_t *f, int rid, int seed_row, int seed_col)
{
    int id = new_seg(f);
    seg_t *s = &f->seg[id];
    s->rid = rid;
    int valid = 1;
    for (int i = seed_row, ic = 0; ic < WIN && i < f->h && valid; ++i, ++ic) {
        for (int j = seed_col, jc = 0; jc < WIN && j < f->w; ++j, ++jc) {
            double x = 0, y = 0, z = 10000;
            if (!cloud_get(f, i, j, &x, &y, &z)) { valid = 0; break; }
            double xn = 0, yn = 0, zn = 10000;
            if (j + 1 < f->w && cloud_get(f, i, j + 1, &xn, &yn, &zn) && fabs(z - zn) > P_DEPTH_ALPHA * fabs(z) + P_DEPTH_CHANGE_TOL) { valid = 0; break; }
            if (i + 1 < f->h && cloud_get(f, i + 1, j, &xn, &yn, &zn) && fabs(z - zn) > P_DEPTH_ALPHA * fabs(z) + P_DEPTH_CHANGE_TOL) { valid = 0; break; }
            stats_t *t = &s->st;
            t->sx += x; t->sy += y; t->sz += z;
            t->sxx += x * x; t->syy += y * y; t->szz += z * z;
            t->sxy += x * y; t->syz += y * z; t->sxz += x * z;
            ++t->N;
        }
    }
    if (valid) { s->nouse = 0; s->N = s->st.N; }
    else { s->N = 0; memset(&s->st, 0, sizeof(s->st)); s->nouse = 1; }
    if (s->N < 4) { s->mse = s->curvature = NAN; }
    else stats_compute(&s->st, s->center, s->normal, &s->mse, &s->curvature);
    return id;
}

/* ahCluster (AHCPlaneFitter.hpp:983-1189) */
static void ah_cluster(fitter_t *f)
{
    int step = 0;
    while (f->nheap > 0 && step <= MAX_STEP) {
        int p = heap_pop(f);
        if (f->seg[p].nouse) continue;
        int cand = -1, cand_nb = -1;
        seg_t *sp = &f->seg[p];
        for (int k = 0; k < sp->nnb; k++) {
            int nb = sp->nbs[k];
            if (normal_similarity(sp, &f->seg[nb]) < cos(deg2rad(60.0))) continue;    /* T_ang(P_MERGING) */
            int m = new_seg(f);
            sp = &f->seg[p];                          /* realloc may have moved */
            seg_t *sm = &f->seg[m], *sn = &f->seg[nb];
            /* PlaneSeg(pa, pb): AHCPlaneSeg.hpp:293-316 */
            sm->st.sx = sp->st.sx + sn->st.sx; sm->st.sy = sp->st.sy + sn->st.sy; sm->st.sz = sp->st.sz + sn->st.sz;
            sm->st.sxx = sp->st.sxx + sn->st.sxx; sm->st.syy = sp->st.syy + sn->st.syy; sm->st.szz = sp->st.szz + sn->st.szz;
            sm->st.sxy = sp->st.sxy + sn->st.sxy; sm->st.syz = sp->st.syz + sn->st.syz; sm->st.sxz = sp->st.sxz + sn->st.sxz;
            sm->st.N = sp->st.N + sn->st.N;
            sm->nouse = 0;
            sm->rid = sp->N >= sn->N ? sp->rid : sn->rid;
            sm->N = sm->st.N;
            stats_compute(&sm->st, sm->center, sm->normal, &sm->mse, &sm->curvature);
            if (cand < 0 || f->seg[cand].mse > sm->mse ||
                (f->seg[cand].mse == sm->mse && (double)f->seg[cand].N < sm->mse)) {   /* sic: N < mse */
                cand = m; cand_nb = nb;
            }
        }
        sp = &f->seg[p];
        if (cand >= 0 && f->seg[cand].mse < T_mse_merge(f->seg[cand].center[2])) {
            /* candidates are created in ascending id order; the queue and the graph only ever see
             * the accepted one, the rejected temporaries stay isolated (nouse irrelevant) */
            heap_push(f, cand);
            /* mergeNbsFrom (AHCPlaneSeg.hpp:379-409) */
            seg_t *sm = &f->seg[cand], *sa = &f->seg[p], *sb = &f->seg[cand_nb];
            ds_union(f, sa->rid, sb->rid);
            for (int i = 0; i < sa->nnb; i++) nb_insert(sm, sa->nbs[i]);
            sa = &f->seg[p]; sb = &f->seg[cand_nb];
            for (int i = 0; i < sb->nnb; i++) nb_insert(sm, sb->nbs[i]);
            nb_erase(sm, p); nb_erase(sm, cand_nb);
            disconnect_all(f, p); disconnect_all(f, cand_nb);
            sm = &f->seg[cand];
            for (int i = 0; i < sm->nnb; i++) nb_insert(&f->seg[sm->nbs[i]], cand);
            f->seg[p].nouse = f->seg[cand_nb].nouse = 1;
        } else {
            if (sp->N >= MIN_SUPPORT) f->extracted[f->nextracted++] = p;
            disconnect_all(f, p);
        }
        ++step;
    }
    while (f->nheap > 0) {
        int p = heap_pop(f);
        if (f->seg[p].N >= MIN_SUPPORT) f->extracted[f->nextracted++] = p;
        disconnect_all(f, p);
    }
    /* std::sort by N descending; ties -> extraction order (insertion sort is stable) */
    for (int i = 1; i < f->nextracted; i++) {
        int v = f->extracted[i], j = i - 1;
        while (j >= 0 && f->seg[f->extracted[j]].N < f->seg[v].N) { f->extracted[j + 1] = f->extracted[j]; j--; }
        f->extracted[j + 1] = v;
    }
}

static int valid4(int i, int j, int H, int W, int nbs[4])
{
    const int id = i * W + j; int c = 0;
    if (j > 0) nbs[c++] = id - 1;
    if (j < W - 1) nbs[c++] = id + 1;
    if (i > 0) nbs[c++] = id - W;
    if (i < H - 1) nbs[c++] = id + W;
    return c;
}

int orc_peac_run(const uint16_t *depth, int w, int h, int stride_bytes,
                 float fx, float fy, float cx, float cy, float depth_factor,
                 int32_t *labels, orc_plane *planes, int cap, int *nplanes)
{
    fitter_t F; memset(&F, 0, sizeof(F));
    fitter_t *f = &F;
    f->w = w; f->h = h; f->Nw = w / WIN; f->Nh = h / WIN;
    const int npix = w * h, nblk = f->Nw * f->Nh;
    double *X = (double *)malloc(sizeof(double) * npix), *Y = (double *)malloc(sizeof(double) * npix), *Z = (double *)malloc(sizeof(double) * npix);
    /* readDepthImage (PlaneExtractor.cpp:42-56): double z = (double)u16 * (float)scale, K as float */
    for (int i = 0; i < h; i++) {
        const uint16_t *row = (const uint16_t *)((const uint8_t *)depth + (size_t)i * stride_bytes);
        for (int j = 0; j < w; j++) {
            double z = (double)row[j] * depth_factor;
            double x = ((double)j - cx) * z / fx;
            double y = ((double)i - cy) * z / fy;
            X[i * w + j] = x; Y[i * w + j] = y; Z[i * w + j] = z;
        }
    }
    f->X = X; f->Y = Y; f->Z = Z;
    f->parent = (int *)malloc(sizeof(int) * nblk); f->dsize = (int *)malloc(sizeof(int) * nblk);
    for (int i = 0; i < nblk; i++) { f->parent[i] = i; f->dsize[i] = 1; }
    f->extracted = (int *)malloc(sizeof(int) * (nblk + 16));
    int *G = (int *)malloc(sizeof(int) * nblk);

    /* ---- initGraph ---- */
    for (int i = 0; i < f->Nh; i++)
        for (int j = 0; j < f->Nw; j++) {
            int id = init_block(f, i * f->Nw + j, i * WIN, j * WIN);
            seg_t *s = &f->seg[id];
            if (s->mse < T_mse_init(s->center[2]) && !s->nouse) { G[i * f->Nw + j] = id; heap_push(f, id); }
            else G[i * f->Nw + j] = -1;
        }
    const int Nw = f->Nw, Nh = f->Nh;
    for (int i = 0; i < Nh; ++i)
        for (int j = 1; j < Nw; j += 2) {
            const int c = i * Nw + j;
            if (G[c - 1] < 0) { --j; continue; }
            if (G[c] < 0) continue;
            if (j < Nw - 1 && G[c + 1] < 0) { ++j; continue; }
            const double th = T_ang_init(f->seg[G[c]].center[2]);
            if ((j < Nw - 1 && normal_similarity(&f->seg[G[c - 1]], &f->seg[G[c + 1]]) >= th) ||
                (j == Nw - 1 && normal_similarity(&f->seg[G[c]], &f->seg[G[c - 1]]) >= th)) {
                connect_seg(f, G[c], G[c - 1]);
                if (j < Nw - 1) connect_seg(f, G[c], G[c + 1]);
            } else --j;
        }
    for (int j = 0; j < Nw; ++j)
        for (int i = 1; i < Nh; i += 2) {
            const int c = i * Nw + j;
            if (G[c - Nw] < 0) { --i; continue; }
            if (G[c] < 0) continue;
            if (i < Nh - 1 && G[c + Nw] < 0) { ++i; continue; }
            const double th = T_ang_init(f->seg[G[c]].center[2]);
            if ((i < Nh - 1 && normal_similarity(&f->seg[G[c - Nw]], &f->seg[G[c + Nw]]) >= th) ||
                (i == Nh - 1 && normal_similarity(&f->seg[G[c]], &f->seg[G[c - Nw]]) >= th)) {
                connect_seg(f, G[c], G[c - Nw]);
                if (i < Nh - 1) connect_seg(f, G[c], G[c + Nw]);
            } else --i;
        }

#ifdef ORC_PEAC_CLUSTER_HOOK                /* tools/ahc_spec_sim.c: the same loop with instrumentation (this file is #included there) */
    ORC_PEAC_CLUSTER_HOOK(f);
#else
    ah_cluster(f);
#endif

    /* ---- refineDetails ---- */
    const int nold = f->nextracted;
    int *oldp = (int *)malloc(sizeof(int) * (nold + 1));
    memcpy(oldp, f->extracted, sizeof(int) * nold);
    int *isvalid = (int *)calloc(nold + 1, sizeof(int));
    int *blkMap = (int *)malloc(sizeof(int) * nblk);
    int *member = labels;
    for (int i = 0; i < npix; i++) member[i] = -1;
    /* rfQueue: pairs (pixidx, plid) */
    int capq = npix * 4 + 1024, nq = 0;
    int *qpix = (int *)malloc(sizeof(int) * capq), *qpl = (int *)malloc(sizeof(int) * capq);
#define QPUSH(px, pl) do { if (nq == capq) { capq *= 2; qpix = (int *)realloc(qpix, sizeof(int) * capq); qpl = (int *)realloc(qpl, sizeof(int) * capq); } qpix[nq] = (px); qpl[nq] = (pl); nq++; } while (0)
    /* rid2plid */
    int *rid2plid = (int *)malloc(sizeof(int) * nblk);
    for (int i = 0; i < nblk; i++) rid2plid[i] = 0;       /* std::map::operator[] default */
    for (int p = 0; p < nold; p++) rid2plid[f->seg[oldp[p]].rid] = p;
    for (int i = 0, blkid = 0; i < Nh; ++i)
        for (int j = 0; j < Nw; ++j, ++blkid) {
            const int setid = ds_find(f, blkid);
            const int setSize = f->dsize[setid] * WIN * WIN;
            if (setSize >= MIN_SUPPORT) {
                int nbs[4]; const int nn = valid4(i, j, Nh, Nw, nbs);
                int same = 1;
                for (int k = 0; k < nn; k++) if (ds_find(f, nbs[k]) != setid) { same = 0; break; }   /* ERODE_ALL_BORDER */
                const int plid = rid2plid[setid];
                if (same) {
                    blkMap[blkid] = plid;
                    for (int y = i * WIN; y < (i + 1) * WIN; y++) for (int x = j * WIN; x < (j + 1) * WIN; x++) member[y * w + x] = plid;
                    isvalid[plid] = 1;
                } else blkMap[blkid] = -1;
            } else blkMap[blkid] = -1;
            if (blkMap[blkid] < 0) {
                if (i > 0 && blkMap[blkid - Nw] >= 0) {
                    const int u = blkMap[blkid - Nw], sp = (i * WIN - 1) * w + j * WIN;
                    for (int k = 1; k < WIN; ++k) QPUSH(sp + k, u);
                }
                if (j > 0 && blkMap[blkid - 1] >= 0) {
                    const int l = blkMap[blkid - 1], sp = (i * WIN) * w + j * WIN - 1;
                    for (int k = 0; k < WIN - 1; ++k) QPUSH(sp + k * w, l);
                }
            } else {
                const int plid = blkMap[blkid];
                if (i > 0 && blkMap[blkid - Nw] != plid) {
                    const int sp = (i * WIN) * w + j * WIN;
                    for (int k = 0; k < WIN - 1; ++k) QPUSH(sp + k, plid);
                }
                if (j > 0 && blkMap[blkid - 1] != plid) {
                    const int sp = (i * WIN) * w + j * WIN;
                    for (int k = 1; k < WIN; ++k) QPUSH(sp + k * w, plid);
                }
            }
        }
    /* ---- floodFill (AHCPlaneFitter.hpp:428-476) ---- */
    float *distMap = (float *)malloc(sizeof(float) * npix);
    for (int i = 0; i < npix; i++) distMap[i] = FLT_MAX;
    const double th_refine = cos(deg2rad(30.0));
#ifdef ORC_PEAC_FLOOD_STATS                 /* tools/flood_gen_sim.c: generations of the FIFO (entries pushed by the generation before) */
    int gen_end = nq, ngen = 0; ORC_PEAC_FLOOD_STATS(0, 0, nq);
#endif
    for (int k = 0; k < nq; ++k) {
#ifdef ORC_PEAC_FLOOD_STATS
        if (k == gen_end) { ++ngen; ORC_PEAC_FLOOD_STATS(ngen, k, nq); gen_end = nq; }
#endif
        const int sIdx = qpix[k], seedy = sIdx / w, seedx = sIdx - seedy * w, plid = qpl[k];
        const seg_t *pl = &f->seg[oldp[plid]];
        int nbs[4]; const int nn = valid4(seedy, seedx, h, w, nbs);
        for (int it = 0; it < nn; ++it) {
            const int cIdx = nbs[it];
            int *trail = &member[cIdx];
            if (*trail <= -6) continue;
            if (*trail >= 0 && *trail == plid) continue;
            const int cy_ = cIdx / w, cx_ = cIdx - cy_ * w;
            const int by = cy_ / WIN, bx = cx_ / WIN;
            const int blkid = (by < Nh && bx < Nw) ? by * Nw + bx : -1;
            if (blkid >= 0 && blkMap[blkid] >= 0) continue;
            double pt[3] = { 0, 0, 0 };
            float cdist = -1;
            int ok = cloud_get(f, cy_, cx_, &pt[0], &pt[1], &pt[2]);
            if (ok) {
                double sd = pl->normal[0] * (pt[0] - pl->center[0]) + pl->normal[1] * (pt[1] - pl->center[1]) + pl->normal[2] * (pt[2] - pl->center[2]);
                cdist = (float)fabs(sd);
                ok = ((double)cdist * (double)cdist) < 9 * pl->mse + 1e-5;
            }
            if (ok) {
                if (*trail >= 0) {
                    seg_t *npl = &f->seg[oldp[*trail]];
                    if (normal_similarity(pl, npl) >= th_refine) connect_seg(f, oldp[*trail], oldp[plid]);
                }
                if (cdist < distMap[cIdx]) { *trail = plid; distMap[cIdx] = cdist; QPUSH(cIdx, plid); }
                else if (*trail < 0) *trail -= 1;
            } else {
                if (*trail < 0) *trail -= 1;
            }
        }
    }
    /* ---- one last merge round over the still-valid planes ---- */
    f->nextracted = 0; f->nheap = 0;
    for (int p = 0; p < nold; p++) if (isvalid[p]) heap_push(f, oldp[p]);
    ah_cluster(f);
    int *plidmap = (int *)malloc(sizeof(int) * (nold + 1));
    for (int p = 0; p < nold; p++) {
        plidmap[p] = -1;
        if (!isvalid[p]) continue;
        const int np_rid = ds_find(f, f->seg[oldp[p]].rid);
        for (int j = 0; j < f->nextracted; j++) if (f->seg[f->extracted[j]].rid == np_rid) { plidmap[p] = j; break; }
    }
    for (int i = 0; i < npix; i++) {
        int pl = member[i];
        member[i] = (pl >= 0 && plidmap[pl] >= 0) ? plidmap[pl] : -1;
    }
    int nout = f->nextracted < cap ? f->nextracted : cap;
    for (int j = 0; j < nout; j++) {
        const seg_t *s = &f->seg[f->extracted[j]];
        memcpy(planes[j].normal, s->normal, sizeof(double) * 3);
        memcpy(planes[j].center, s->center, sizeof(double) * 3);
        planes[j].mse = s->mse; planes[j].n_points = s->N; planes[j].rid = s->rid;
    }
    *nplanes = f->nextracted;
    for (int i = 0; i < f->nseg; i++) free(f->seg[i].nbs);
    free(f->seg); free(f->parent); free(f->dsize); free(f->heap); free(f->extracted); free(G);
    free(oldp); free(isvalid); free(blkMap); free(qpix); free(qpl); free(rid2plid); free(distMap); free(plidmap);
    free(X); free(Y); free(Z);
    return 0;
}

/* threshold accessors for the known-answer tests */
double orc_peac_T_mse_init(double z) { return T_mse_init(z); }
double orc_peac_T_ang_init(double z) { return T_ang_init(z); }
double orc_peac_T_dz(double z) { return P_DEPTH_ALPHA * fabs(z) + P_DEPTH_CHANGE_TOL; }
double orc_peac_T_mse_merge(double z) { return T_mse_merge(z); }

/* The union-find the fitter uses, behind test hooks: tests/test_ref_pins.py runs random union sequences through
 * these and through the reference's own include/peac/DisjointSet.hpp (compiled as it stands into oracle/_ref). */
typedef struct { fitter_t f; } orc_ds;
void *orc_ds_create(int n)
{
    orc_ds *d = (orc_ds *)calloc(1, sizeof(orc_ds));
    d->f.parent = (int *)malloc(sizeof(int) * (n > 0 ? n : 1)); d->f.dsize = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { d->f.parent[i] = i; d->f.dsize[i] = 1; }
    return d;
}
int  orc_ds_union(void *d, int x, int y) { return ds_union(&((orc_ds *)d)->f, x, y); }
int  orc_ds_find(void *d, int x) { return ds_find(&((orc_ds *)d)->f, x); }
int  orc_ds_set_size(void *d, int x) { orc_ds *q = (orc_ds *)d; return q->f.dsize[ds_find(&q->f, x)]; }
void orc_ds_free(void *d) { orc_ds *q = (orc_ds *)d; free(q->f.parent); free(q->f.dsize); free(q); }

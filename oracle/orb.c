/*
 * orb.c -- ORACLE (test infrastructure only; see oracle.h).  Parity unpinned.
 *
 * CPU restatement of ORB_SLAM2::ORBextractor as used by the reference:
 *   constructor / tables      src/ORBextractor.cc:408-468
 *   operator()                src/ORBextractor.cc:1041-1103
 *   ComputePyramid            src/ORBextractor.cc:1105-1130
 *   ComputeKeyPointsOctTree   src/ORBextractor.cc:763-851
 *   DistributeOctTree         src/ORBextractor.cc:537-761 (+ DivideNode 479-535)
 *   IC_Angle                  src/ORBextractor.cc:75-102
 *   computeOrbDescriptor      src/ORBextractor.cc:106-145
 *
 * Determinism rules where the reference is address/compiler dependent:
 *   - std::sort of pair<int,ExtractorNode*> (ORBextractor.cc:682): ties on size are broken by
 *     node creation sequence number (what a bump allocator would give).
 *   - x*b + y*a in the descriptor is evaluated without FMA contraction (-ffp-contract=off).
 *   - cos/sin of a float argument: evaluated in double and rounded to float.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31
#define HALF_PATCH_SIZE 15
#define EDGE_THRESHOLD 19
#define MAXLEV 16

static const int8_t g_pattern[1024] = {
#include "orb_pattern.inc"
};
const int8_t *orc_orb_pattern(void) { return g_pattern; }

typedef struct { int w, h, stride; uint8_t *img, *blur; } level_t;

struct orc_orb {
    orc_orb_params p;
    float scale[MAXLEV], inv_scale[MAXLEV];
    int nfeat[MAXLEV];
    int umax[HALF_PATCH_SIZE + 1];
    level_t lev[MAXLEV];
    int *cand[MAXLEV]; int ncand[MAXLEV], capcand[MAXLEV];
    int grid[MAXLEV][4];
    int nkept[MAXLEV];
};

orc_orb *orc_orb_create(const orc_orb_params *p)
{
    if (p->nlevels < 1 || p->nlevels > MAXLEV) return NULL;
    orc_orb *o = (orc_orb *)calloc(1, sizeof(*o));
    o->p = *p;
    /* ORBextractor.cc:413-428 */
    o->scale[0] = 1.0f;
    for (int i = 1; i < p->nlevels; i++) o->scale[i] = o->scale[i - 1] * p->scale_factor;
    for (int i = 0; i < p->nlevels; i++) o->inv_scale[i] = 1.0f / o->scale[i];
    /* ORBextractor.cc:432-444 */
    float factor = 1.0f / p->scale_factor;
    float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p->nlevels));
    int sum = 0;
    for (int l = 0; l < p->nlevels - 1; l++) {
        o->nfeat[l] = orc_cvround_f(nd);
        sum += o->nfeat[l];
        nd *= factor;
    }
    o->nfeat[p->nlevels - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    /* ORBextractor.cc:452-467 */
    int v, v0;
    int vmax = (int)floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) o->umax[v] = orc_cvround_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    return o;
}

void orc_orb_destroy(orc_orb *o)
{
    if (!o) return;
    for (int l = 0; l < MAXLEV; l++) { free(o->lev[l].img); free(o->lev[l].blur); free(o->cand[l]); }
    free(o);
}

int orc_orb_umax(const orc_orb *o, int *out) { memcpy(out, o->umax, sizeof(o->umax)); return 16; }
int orc_orb_features_per_level(const orc_orb *o, int *out)
{ memcpy(out, o->nfeat, sizeof(int) * o->p.nlevels); return o->p.nlevels; }
float orc_orb_scale(const orc_orb *o, int l) { return o->scale[l]; }
int orc_orb_level(const orc_orb *o, int l, int *w, int *h, int *stride, const uint8_t **data)
{ *w = o->lev[l].w; *h = o->lev[l].h; *stride = o->lev[l].stride; *data = o->lev[l].img; return 0; }
int orc_orb_level_blurred(const orc_orb *o, int l, const uint8_t **data, int *stride)
{ *data = o->lev[l].blur; *stride = o->lev[l].stride; return o->lev[l].blur ? 0 : -1; }
int orc_orb_grid(const orc_orb *o, int l, int *nc, int *nr, int *wc, int *hc)
{ *nc = o->grid[l][0]; *nr = o->grid[l][1]; *wc = o->grid[l][2]; *hc = o->grid[l][3]; return 0; }
int orc_orb_candidates(const orc_orb *o, int l, const int **xys) { *xys = o->cand[l]; return o->ncand[l]; }
int orc_orb_level_count(const orc_orb *o, int l) { return o->nkept[l]; }

/* ---- ComputePyramid (ORBextractor.cc:1105-1130).  The 19-px REFLECT_101 apron that the
 * reference materialises is never read inside the image-dependent steps except by the blur,
 * where reflecting indices is identical (apron == REFLECT_101 of the level). ---- */
static void compute_pyramid(orc_orb *o, const uint8_t *gray, int w, int h, int stride)
{
    for (int l = 0; l < o->p.nlevels; l++) {
        level_t *L = &o->lev[l];
        float s = o->inv_scale[l];
        int lw = orc_cvround_f((float)w * s), lh = orc_cvround_f((float)h * s);
        if (L->w != lw || L->h != lh) {
            free(L->img); free(L->blur);
            L->w = lw; L->h = lh; L->stride = lw;
            L->img = (uint8_t *)malloc((size_t)lw * lh);
            L->blur = (uint8_t *)malloc((size_t)lw * lh);
        }
        if (l == 0) {
            for (int y = 0; y < h; y++) memcpy(L->img + (size_t)y * lw, gray + (size_t)y * stride, w);
        } else {
            level_t *P = &o->lev[l - 1];
            orc_resize_linear_u8(P->img, P->w, P->h, P->stride, L->img, lw, lh, L->stride);
        }
    }
}

/* ---- quadtree (ORBextractor.cc:479-761) ---- */
typedef struct {
    int ulx, uly, brx, bry;
    int koff, nk;       /* key-index list in the pool */
    int nomore;
    int prev, next;     /* std::list links */
    int alive;
} qnode;

typedef struct {
    qnode *nd; int nn, capn;
    int *pool; int npool, cappool;
    int head, tail, size;
} qtree;

static int qt_new_node(qtree *t)
{
    if (t->nn == t->capn) { t->capn = t->capn * 2 + 64; t->nd = (qnode *)realloc(t->nd, sizeof(qnode) * t->capn); }
    memset(&t->nd[t->nn], 0, sizeof(qnode));
    t->nd[t->nn].prev = t->nd[t->nn].next = -1;
    return t->nn++;
}
static int qt_alloc_keys(qtree *t, int n)
{
    if (t->npool + n > t->cappool) { t->cappool = (t->npool + n) * 2 + 1024; t->pool = (int *)realloc(t->pool, sizeof(int) * t->cappool); }
    int off = t->npool; t->npool += n; return off;
}
static void qt_push_front(qtree *t, int i)
{
    t->nd[i].prev = -1; t->nd[i].next = t->head; t->nd[i].alive = 1;
    if (t->head >= 0) t->nd[t->head].prev = i; else t->tail = i;
    t->head = i; t->size++;
}
static void qt_push_back(qtree *t, int i)
{
    t->nd[i].next = -1; t->nd[i].prev = t->tail; t->nd[i].alive = 1;
    if (t->tail >= 0) t->nd[t->tail].next = i; else t->head = i;
    t->tail = i; t->size++;
}
static int qt_erase(qtree *t, int i)   /* returns next */
{
    int p = t->nd[i].prev, n = t->nd[i].next;
    if (p >= 0) t->nd[p].next = n; else t->head = n;
    if (n >= 0) t->nd[n].prev = p; else t->tail = p;
    t->nd[i].alive = 0; t->size--;
    return n;
}

/* DivideNode: creates up to 4 children records (not yet linked); child[k] = -1 if it would be
 * empty (the reference constructs it and then drops it). Keys keep their relative order. */
static void qt_divide(qtree *t, int ni, const int *xys, int child[4])
{
    qnode n = t->nd[ni];
    const int halfX = (int)ceilf((float)(n.brx - n.ulx) / 2);
    const int halfY = (int)ceilf((float)(n.bry - n.uly) / 2);
    const int mx = n.ulx + halfX, my = n.uly + halfY;
    int cnt[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < n.nk; i++) {
        int k = t->pool[n.koff + i];
        float x = (float)xys[3 * k], y = (float)xys[3 * k + 1];
        int c = (x < (float)mx) ? ((y < (float)my) ? 0 : 2) : ((y < (float)my) ? 1 : 3);
        cnt[c]++;
    }
    int off[4], fill[4] = { 0, 0, 0, 0 };
    for (int c = 0; c < 4; c++) off[c] = qt_alloc_keys(t, cnt[c]);
    for (int i = 0; i < n.nk; i++) {
        int k = t->pool[n.koff + i];
        float x = (float)xys[3 * k], y = (float)xys[3 * k + 1];
        int c = (x < (float)mx) ? ((y < (float)my) ? 0 : 2) : ((y < (float)my) ? 1 : 3);
        t->pool[off[c] + fill[c]++] = k;
    }
    const int bx[4][4] = { { n.ulx, n.uly, mx, my }, { mx, n.uly, n.brx, my },
                           { n.ulx, my, mx, n.bry }, { mx, my, n.brx, n.bry } };
    for (int c = 0; c < 4; c++) {
        if (cnt[c] == 0) { child[c] = -1; continue; }
        int ci = qt_new_node(t);
        qnode *q = &t->nd[ci];
        q->ulx = bx[c][0]; q->uly = bx[c][1]; q->brx = bx[c][2]; q->bry = bx[c][3];
        q->koff = off[c]; q->nk = cnt[c]; q->nomore = (cnt[c] == 1);
        child[c] = ci;
    }
}

typedef struct { int size, node; } szptr;
static int cmp_szptr(const void *a, const void *b)
{
    const szptr *x = (const szptr *)a, *y = (const szptr *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->node < y->node ? -1 : (x->node > y->node);   /* creation sequence (H2 rule) */
}

/* returns number of selected candidates; sel[] receives candidate indices in list order */
static int distribute_octree(const int *xys, int nc, int minX, int maxX, int minY, int maxY,
                             int N, int *sel)
{
    qtree t; memset(&t, 0, sizeof(t)); t.head = t.tail = -1;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) return 0;
    const float hX = (float)(maxX - minX) / nIni;
    int *ini = (int *)malloc(sizeof(int) * nIni);
    int *cnt = (int *)calloc(nIni, sizeof(int));
    for (int i = 0; i < nc; i++) cnt[(int)((float)xys[3 * i] / hX)]++;
    for (int i = 0; i < nIni; i++) {
        int ni = qt_new_node(&t);
        qnode *q = &t.nd[ni];
        q->ulx = (int)(hX * (float)i); q->brx = (int)(hX * (float)(i + 1));
        q->uly = 0; q->bry = maxY - minY;
        q->koff = qt_alloc_keys(&t, cnt[i]); q->nk = 0;
        ini[i] = ni;
        qt_push_back(&t, ni);
    }
    for (int i = 0; i < nc; i++) {
        qnode *q = &t.nd[ini[(int)((float)xys[3 * i] / hX)]];
        t.pool[q->koff + q->nk++] = i;
    }
    for (int it = t.head; it >= 0;) {
        if (t.nd[it].nk == 1) { t.nd[it].nomore = 1; it = t.nd[it].next; }
        else if (t.nd[it].nk == 0) it = qt_erase(&t, it);
        else it = t.nd[it].next;
    }
    szptr *vs = (szptr *)malloc(sizeof(szptr) * (size_t)(nc + 8) * 4), *vp = (szptr *)malloc(sizeof(szptr) * (size_t)(nc + 8) * 4);
    int nvs = 0, finish = 0;
    while (!finish) {
        int prevSize = t.size, nToExpand = 0;
        nvs = 0;
        for (int it = t.head; it >= 0;) {
            if (t.nd[it].nomore) { it = t.nd[it].next; continue; }
            int ch[4];
            qt_divide(&t, it, xys, ch);
            for (int c = 0; c < 4; c++) {
                if (ch[c] < 0) continue;
                qt_push_front(&t, ch[c]);
                if (t.nd[ch[c]].nk > 1) { nToExpand++; vs[nvs].size = t.nd[ch[c]].nk; vs[nvs].node = ch[c]; nvs++; }
            }
            it = qt_erase(&t, it);
        }
        if (t.size >= N || t.size == prevSize) {
            finish = 1;
        } else if (t.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = t.size;
                int nvp = nvs;
                memcpy(vp, vs, sizeof(szptr) * nvs);
                nvs = 0;
                qsort(vp, nvp, sizeof(szptr), cmp_szptr);
                for (int j = nvp - 1; j >= 0; j--) {
                    int ch[4];
                    qt_divide(&t, vp[j].node, xys, ch);
                    for (int c = 0; c < 4; c++) {
                        if (ch[c] < 0) continue;
                        qt_push_front(&t, ch[c]);
                        if (t.nd[ch[c]].nk > 1) { vs[nvs].size = t.nd[ch[c]].nk; vs[nvs].node = ch[c]; nvs++; }
                    }
                    qt_erase(&t, vp[j].node);
                    if (t.size >= N) break;
                }
                if (t.size >= N || t.size == prevSize) finish = 1;
            }
        }
    }
    int ns = 0;
    for (int it = t.head; it >= 0; it = t.nd[it].next) {
        const qnode *q = &t.nd[it];
        int best = t.pool[q->koff];
        float maxr = (float)xys[3 * best + 2];
        for (int k = 1; k < q->nk; k++) {
            int c = t.pool[q->koff + k];
            if ((float)xys[3 * c + 2] > maxr) { best = c; maxr = (float)xys[3 * c + 2]; }
        }
        sel[ns++] = best;
    }
    free(ini); free(cnt); free(vs); free(vp); free(t.nd); free(t.pool);
    return ns;
}

/* IC_Angle (ORBextractor.cc:75-102) */
static float ic_angle(const uint8_t *img, int stride, float px, float py, const int *umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)orc_cvround_f(py) * stride + orc_cvround_f(px);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor (ORBextractor.cc:106-145); `img` is the blurred level, reads are
 * REFLECT_101-safe by construction (keypoints are >=16 px from the border, |offset| <= 15+1;
 * we reflect anyway so that tiny images cannot read out of bounds). */
static void orb_descriptor(const level_t *L, float kx, float ky, float angle_deg, uint8_t *desc)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = angle_deg * factorPI;
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    int cx = orc_cvround_f(kx), cy = orc_cvround_f(ky);
    const int8_t *pat = g_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int t[2];
            for (int e = 0; e < 2; e++) {
                float x = (float)pat[4 * j + 2 * e], y = (float)pat[4 * j + 2 * e + 1];
                int ry = orc_cvround_f(x * b + y * a), rx = orc_cvround_f(x * a - y * b);
                int yy = cy + ry, xx = cx + rx;
                if (yy < 0) yy = -yy; if (yy >= L->h) yy = 2 * (L->h - 1) - yy;
                if (xx < 0) xx = -xx; if (xx >= L->w) xx = 2 * (L->w - 1) - xx;
                t[e] = L->blur[(size_t)yy * L->stride + xx];
            }
            val |= (t[0] < t[1]) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

int orc_orb_extract(orc_orb *o, const uint8_t *gray, int w, int h, int stride,
                    orc_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return 0;            /* ORBextractor.cc:1044 */
    const int nl = o->p.nlevels;
    compute_pyramid(o, gray, w, h, stride);
    int total = 0;
    const float W = 30;
    for (int l = 0; l < nl; l++) {
        level_t *L = &o->lev[l];
        /* ---- ComputeKeyPointsOctTree, ORBextractor.cc:769-827 ---- */
        const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
        const int maxBX = L->w - EDGE_THRESHOLD + 3, maxBY = L->h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        o->ncand[l] = 0; o->nkept[l] = 0;
        o->grid[l][0] = nCols; o->grid[l][1] = nRows; o->grid[l][2] = o->grid[l][3] = 0;
        if (nCols < 1 || nRows < 1) continue;           /* reference would divide by zero */
        const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
        o->grid[l][2] = wCell; o->grid[l][3] = hCell;
        int tmpc[3 * 1024];
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                const uint8_t *view = L->img + (size_t)(int)iniY * L->stride + (int)iniX;
                int vw = (int)maxX - (int)iniX, vh = (int)maxY - (int)iniY;
                int nk = orc_fast9_16(view, L->stride, vw, vh, o->p.ini_th_fast, tmpc, 1024);
                if (nk == 0) nk = orc_fast9_16(view, L->stride, vw, vh, o->p.min_th_fast, tmpc, 1024);
                if (o->ncand[l] + nk > o->capcand[l]) {
                    o->capcand[l] = (o->ncand[l] + nk) * 2 + 1024;
                    o->cand[l] = (int *)realloc(o->cand[l], sizeof(int) * 3 * o->capcand[l]);
                }
                for (int k = 0; k < nk; k++) {
                    int *c = o->cand[l] + 3 * (o->ncand[l] + k);
                    c[0] = tmpc[3 * k] + j * wCell; c[1] = tmpc[3 * k + 1] + i * hCell; c[2] = tmpc[3 * k + 2];
                }
                o->ncand[l] += nk;
            }
        }
        /* ---- DistributeOctTree + border/scale bookkeeping, ORBextractor.cc:829-846 ---- */
        int *sel = (int *)malloc(sizeof(int) * (o->ncand[l] + 1));
        int ns = o->ncand[l] ? distribute_octree(o->cand[l], o->ncand[l], minBX, maxBX, minBY, maxBY, o->nfeat[l], sel) : 0;
        const int scaledPatchSize = (int)(PATCH_SIZE * o->scale[l]);
        for (int k = 0; k < ns && total + k < cap; k++) {
            orc_keypoint *kp = &kps[total + k];
            const int *c = o->cand[l] + 3 * sel[k];
            kp->x = (float)c[0] + minBX; kp->y = (float)c[1] + minBY;
            kp->response = (float)c[2]; kp->octave = l; kp->size = (float)scaledPatchSize;
            kp->class_id = -1; kp->angle = -1;
        }
        free(sel);
        if (total + ns > cap) ns = cap - total;
        o->nkept[l] = ns;
        total += ns;
    }
    /* orientations (ORBextractor.cc:849-850) */
    int off = 0;
    for (int l = 0; l < nl; l++) {
        level_t *L = &o->lev[l];
        for (int k = 0; k < o->nkept[l]; k++)
            kps[off + k].angle = ic_angle(L->img, L->stride, kps[off + k].x, kps[off + k].y, o->umax);
        off += o->nkept[l];
    }
    /* blur + descriptors + rescale (ORBextractor.cc:1073-1102) */
    off = 0;
    for (int l = 0; l < nl; l++) {
        level_t *L = &o->lev[l];
        int nk = o->nkept[l];
        if (nk == 0) continue;
        orc_gaussian_blur_u8(L->img, L->w, L->h, L->stride, L->blur, L->stride, 7, 2.0);
        for (int k = 0; k < nk; k++)
            orb_descriptor(L, kps[off + k].x, kps[off + k].y, kps[off + k].angle, desc + (size_t)(off + k) * 32);
        if (l != 0) {
            float s = o->scale[l];
            for (int k = 0; k < nk; k++) { kps[off + k].x *= s; kps[off + k].y *= s; }
        }
        off += nk;
    }
    *n = total;
    return 0;
}

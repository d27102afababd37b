/* ahc_spec_sim.c -- how many SEQUENTIAL rounds does ahCluster need if the K smallest heap entries are evaluated together and
 * the longest prefix of mutually independent ones is committed per round?  (Design study for k_peac_cluster; not product code.)
 *
 * A head p_j (j-th smallest heap entry) may commit in the same round as the heads before it iff
 *   (1) S_j = N[p_j] u N[nb_j] (closed neighbourhoods of the node and of its merge partner) is disjoint from every earlier
 *       committed S_i  -> its evaluation read nothing an earlier commit changed and its list edits touch other lists;
 *   (2) no node created by an earlier commit of the round precedes p_j in heap order (m_i < mse(p_j)).
 * The round ends at the first head that fails (prefix rule keeps creation ids in sequential order).
 * The simulation checks itself: after predicting a prefix it executes that many real sequential steps and compares the
 * popped nodes and decisions with the prediction.
 *
 *   gcc -O2 -o /tmp/ahc_spec_sim tools/ahc_spec_sim.c -lm && /tmp/ahc_spec_sim depth.u16 640 480 8
 */
#include <stdio.h>
#define ORC_PEAC_CLUSTER_HOOK spec_cluster
struct fitter_s;
static void spec_cluster(void *f);
#include "../oracle/peac.c"

static int g_K = 8;
static long g_rounds, g_pops, g_hist[65], g_cands, g_passes64, g_passes16, g_stop[3];

typedef struct { int p, nb, merge; double m; int ncand; } head_t;

static void evaluate(fitter_t *f, int p, head_t *h)
{
    seg_t *sp = &f->seg[p];
    h->p = p; h->nb = -1; h->merge = 0; h->m = 0; h->ncand = 0;
    double cm = 0, cc2 = 0; int cN = 0, have = 0;
    for (int k = 0; k < sp->nnb; k++) {
        int nb = sp->nbs[k];
        seg_t *sn = &f->seg[nb];
        if (normal_similarity(sp, sn) < cos(deg2rad(60.0))) continue;
        stats_t st = sp->st;
        st.sx += sn->st.sx; st.sy += sn->st.sy; st.sz += sn->st.sz; st.sxx += sn->st.sxx; st.syy += sn->st.syy; st.szz += sn->st.szz;
        st.sxy += sn->st.sxy; st.syz += sn->st.syz; st.sxz += sn->st.sxz; st.N += sn->st.N;
        double c[3], n[3], mse, curv;
        stats_compute(&st, c, n, &mse, &curv);
        h->ncand++;
        if (!have || cm > mse || (cm == mse && (double)cN < mse)) { have = 1; cm = mse; cN = st.N; cc2 = c[2]; h->nb = nb; }
    }
    if (have && cm < T_mse_merge(cc2)) { h->merge = 1; h->m = cm; }
}

static int in_set(const int *set, int n, int v) { for (int i = 0; i < n; i++) if (set[i] == v) return 1; return 0; }

static void spec_cluster(void *fv)
{
    fitter_t *f = (fitter_t *)fv;
    int *mark = NULL; int markcap = 0;
    while (f->nheap > 0) {
        /* the K smallest live heap entries, in order (copy of the heap) */
        int *hc = (int *)malloc(sizeof(int) * f->nheap); memcpy(hc, f->heap, sizeof(int) * f->nheap);
        int save_n = f->nheap; int *save_h = f->heap;
        f->heap = hc;
        head_t heads[64]; int nh = 0;
        while (nh < g_K && f->nheap > 0) { int p = heap_pop(f); if (f->seg[p].nouse) continue; evaluate(f, p, &heads[nh]); nh++; }
        f->heap = save_h; f->nheap = save_n; free(hc);
        if (nh == 0) { while (f->nheap > 0) heap_pop(f); break; }
        /* prefix rule */
        if (markcap < f->nseg + 64) { markcap = f->nseg * 2 + 1024; mark = (int *)realloc(mark, sizeof(int) * markcap); }
        for (int i = 0; i < f->nseg; i++) mark[i] = 0;
        int L = 0; double minnew = 1e300; long cands = 0;
        for (int j = 0; j < nh; j++) {
            head_t *h = &heads[j];
            int ok = 1;
            if (minnew < f->seg[h->p].mse) { ok = 0; g_stop[0]++; }
            seg_t *sp = &f->seg[h->p];
            if (ok && mark[h->p]) { ok = 0; g_stop[1]++; }
            else if (ok && h->merge && mark[h->nb]) g_stop[2]++;
            for (int k = 0; ok && k < sp->nnb; k++) if (mark[sp->nbs[k]]) ok = 0;
            if (ok && h->merge) { seg_t *sn = &f->seg[h->nb]; if (mark[h->nb]) ok = 0; for (int k = 0; ok && k < sn->nnb; k++) if (mark[sn->nbs[k]]) ok = 0; }
            if (!ok) break;
            mark[h->p] = 1; for (int k = 0; k < sp->nnb; k++) mark[sp->nbs[k]] = 1;
            if (h->merge) { seg_t *sn = &f->seg[h->nb]; mark[h->nb] = 1; for (int k = 0; k < sn->nnb; k++) mark[sn->nbs[k]] = 1; if (h->m < minnew) minnew = h->m; }
            L++;
        }
        for (int j = 0; j < nh; j++) cands += heads[j].ncand;          /* all K heads are evaluated, committed or not */
        g_cands += cands; g_passes64 += (cands + 63) / 64 > 0 ? (cands + 63) / 64 : 1; g_passes16 += (cands + 15) / 16 > 0 ? (cands + 15) / 16 : 1;
        /* execute L real sequential steps and compare */
        for (int j = 0; j < L; j++) {
            int p;
            do { p = heap_pop(f); } while (f->seg[p].nouse);
            if (p != heads[j].p) { fprintf(stderr, "MISPREDICT head %d: popped %d expected %d\n", j, p, heads[j].p); exit(1); }
            head_t now; evaluate(f, p, &now);
            if (now.merge != heads[j].merge || now.nb != heads[j].nb || (now.merge && now.m != heads[j].m)) { fprintf(stderr, "MISPREDICT decision at head %d\n", j); exit(1); }
            /* the real step (copy of ah_cluster's body) */
            if (now.merge) {
                int m = new_seg(f);
                seg_t *sm = &f->seg[m], *sa = &f->seg[p], *sb = &f->seg[now.nb];
                sm->st = sa->st; sm->st.sx += sb->st.sx; sm->st.sy += sb->st.sy; sm->st.sz += sb->st.sz; sm->st.sxx += sb->st.sxx; sm->st.syy += sb->st.syy;
                sm->st.szz += sb->st.szz; sm->st.sxy += sb->st.sxy; sm->st.syz += sb->st.syz; sm->st.sxz += sb->st.sxz; sm->st.N += sb->st.N;
                sm->nouse = 0; sm->rid = sa->N >= sb->N ? sa->rid : sb->rid; sm->N = sm->st.N;
                stats_compute(&sm->st, sm->center, sm->normal, &sm->mse, &sm->curvature);
                heap_push(f, m);
                ds_union(f, sa->rid, sb->rid);
                for (int i = 0; i < sa->nnb; i++) nb_insert(sm, sa->nbs[i]);
                for (int i = 0; i < sb->nnb; i++) nb_insert(sm, sb->nbs[i]);
                nb_erase(sm, p); nb_erase(sm, now.nb);
                disconnect_all(f, p); disconnect_all(f, now.nb);
                for (int i = 0; i < sm->nnb; i++) nb_insert(&f->seg[sm->nbs[i]], m);
                f->seg[p].nouse = f->seg[now.nb].nouse = 1;
            } else {
                if (f->seg[p].N >= MIN_SUPPORT) f->extracted[f->nextracted++] = p;
                disconnect_all(f, p);
            }
            g_pops++;
        }
        g_rounds++; g_hist[L]++;
    }
    for (int i = 1; i < f->nextracted; i++) {
        int v = f->extracted[i], j = i - 1;
        while (j >= 0 && f->seg[f->extracted[j]].N < f->seg[v].N) { f->extracted[j + 1] = f->extracted[j]; j--; }
        f->extracted[j + 1] = v;
    }
    free(mark);
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s depth.u16 w h K\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]); g_K = atoi(argv[4]);
    uint16_t *d = (uint16_t *)malloc((size_t)w * h * 2);
    FILE *fp = fopen(argv[1], "rb"); if (!fp || fread(d, 2, (size_t)w * h, fp) != (size_t)w * h) { fprintf(stderr, "read failed\n"); return 1; }
    fclose(fp);
    int32_t *labels = (int32_t *)malloc((size_t)w * h * 4); orc_plane planes[64]; int np = 0;
    orc_peac_run(d, w, h, w * 2, 535.4f, 539.2f, 320.1f, 247.6f, 1.0f / 5000.0f, labels, planes, 64, &np);
    long long csum = 0; for (int i = 0; i < w * h; i++) csum = csum * 31 + labels[i];
    printf("K=%d pops=%ld rounds=%ld pops/round=%.2f planes=%d labelsum=%lld cands/round=%.1f eig-passes/round: 64 lanes %.2f, 16 lanes %.2f\n", g_K, g_pops, g_rounds,
           (double)g_pops / g_rounds, np, csum, (double)g_cands / g_rounds, (double)g_passes64 / g_rounds, (double)g_passes16 / g_rounds);
    printf("prefix stops: a node created this round precedes the head in queue order %ld, head in an earlier footprint %ld, partner in an earlier footprint %ld (the rest: shared neighbours)\n", g_stop[0], g_stop[1], g_stop[2]);
    printf("prefix-length histogram:"); for (int i = 1; i <= g_K; i++) printf(" %d:%ld", i, g_hist[i]); printf("\n");
    return 0;
}

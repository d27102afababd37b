/* lsd_async_sim.c -- design study AND executable specification (not product code) of the asynchronous multi-region line growing:
 * W workers grow regions side by side, each against the COMMITTED mask as it stood when the worker started (its own claims private),
 * results are committed strictly in seed (raster) order after a validation that makes the outcome the sequential one:
 *   (1) the result's seed is the first available pixel of the committed mask at or after the commit frontier, and
 *   (2) every pixel the region ever claimed (first growth, refine's re-growth) is still available in the committed mask.
 * (An unavailable pixel a worker saw was used by a COMMITTED region: final.  An available pixel it tested and did not take was not
 * aligned with the region's angle at that moment: not taken whatever happens to it.  So (1) + (2) = same decisions as the sequential run.)
 * A result that fails is thrown away; if the frontier's seed has no valid result, it is grown again against the committed mask --
 * by construction valid.  The simulator executes exactly this and asserts that the committed sequence equals the oracle's.
 *
 * Event-driven timing model: a region costs (points / PPR rounds + SEED_OVH) x ROUND, + region2rect / refine passes; validation and commit
 * cost VAL per 64 claimed pixels on the committing worker.  Seed guesses: starts of runs of available pixels behind the last guess,
 * skipping pixels that finished-but-uncommitted results hold (what the device can see in the owner tags).
 *
 *   gcc -O2 -o /tmp/lsd_async_sim tools/lsd_async_sim.c -lm && /tmp/lsd_async_sim gray.u8 640 480 8
 */
#include <stdio.h>
#define orc_lsd_detect orc_lsd_detect_unused
#include "../oracle/lsd.c"
#undef orc_lsd_detect

typedef struct {
    int seed; int n_ever, n_final, n_blocked; int *ever, *fin, *blocked; int big, seg; float s[4];
    double cost, t_start, t_done; int worker; int state;      /* 0 running, 1 done */
} spec_t;

static lsd_t L; static size_t np; static double prec, p_; static unsigned min_reg;
static regpt *reg; static uint8_t *scratch;
static double ROUND = 1.0, PPR = 2.8, SEED_OVH = 1.5; static int TAGS_LIVE = 1, SEE_OLDER = 1;

/* region_grow for a speculating worker: a pixel an OLDER region in flight holds (owner tag) counts as used; the ones that were aligned
 * -- that the sequential run takes if the older region does not keep them -- are recorded: they must be USED at this region's commit */
static const uint8_t *g_held = NULL; static int *g_blocked = NULL; static int g_nblocked = 0, g_capblocked = 0;
static void region_grow_spec(lsd_t *Lx, int sx, int sy, regpt *rg, int *reg_size, double *reg_angle, double prc)
{
    *reg_size = 1;
    rg[0].x = sx; rg[0].y = sy;
    int addr = sx + sy * Lx->w;
    *reg_angle = Lx->angles[addr];
    rg[0].angle = *reg_angle; rg[0].modgrad = Lx->modgrad[addr];
    float sumdx = (float)cos(*reg_angle), sumdy = (float)sin(*reg_angle);
    Lx->used[addr] = USED;
    for (int i = 0; i < *reg_size; ++i)
        for (int xx = rg[i].x - 1; xx <= rg[i].x + 1; ++xx)
            for (int yy = rg[i].y - 1; yy <= rg[i].y + 1; ++yy) {
                int c_addr = xx + yy * Lx->w;
                if ((xx >= 0 && yy >= 0) && (xx < Lx->w && yy < Lx->h) && (Lx->used[c_addr] != USED) && is_aligned(Lx, c_addr, *reg_angle, prc)) {
                    if (g_held && g_held[c_addr]) {
                        if (g_nblocked == g_capblocked) { g_capblocked = g_capblocked ? 2 * g_capblocked : 1024; g_blocked = (int *)realloc(g_blocked, sizeof(int) * g_capblocked); }
                        g_blocked[g_nblocked++] = c_addr; continue;
                    }
                    Lx->used[c_addr] = USED;
                    regpt *rp = &rg[*reg_size];
                    rp->x = xx; rp->y = yy; rp->modgrad = Lx->modgrad[c_addr];
                    const double angle = Lx->angles[c_addr];
                    rp->angle = angle; ++*reg_size;
                    sumdx += cos((float)angle); sumdy += sin((float)angle);
                    *reg_angle = orc_fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                }
            }
}
#define region_grow region_grow_spec

/* refine with the re-growth's claims recorded */
static int refine_rec(lsd_t *Lx, regpt *rg, int *reg_size, double reg_angle, rect_t *rec, int **ever, int *ne, double *cost)
{
    double density = (double)*reg_size / (distd(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density >= 0.7) return 1;
    double xc = (double)rg[0].x, yc = (double)rg[0].y; const double ang_c = rg[0].angle;
    double sum = 0, s_sum = 0; int n = 0;
    *cost += *reg_size / 64.0 + 1;
    for (int i = 0; i < *reg_size; ++i) {
        Lx->used[rg[i].x + rg[i].y * Lx->w] = NOTUSED;
        if (distd(xc, yc, rg[i].x, rg[i].y) < rec->width) { double ang_d = angle_diff_signed(rg[i].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
    }
    double mean_angle = sum / (double)n;
    double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
    region_grow(Lx, rg[0].x, rg[0].y, rg, reg_size, &reg_angle, tau);
    *cost += *reg_size / PPR + SEED_OVH;
    *ever = (int *)realloc(*ever, sizeof(int) * (*ne + *reg_size + 1));
    for (int i = 0; i < *reg_size; i++) (*ever)[(*ne)++] = rg[i].x + rg[i].y * Lx->w;
    if (*reg_size < 2) return 0;
    region2rect(rg, *reg_size, reg_angle, prec, p_, rec);
    *cost += 3 * (*reg_size / 64.0 + 1);
    density = (double)*reg_size / (distd(rec->x1, rec->y1, rec->x2, rec->y2) * rec->width);
    if (density < 0.7) { *cost += 4 * (*reg_size / 64.0 + 1); return reduce_region_radius(Lx, rg, reg_size, reg_angle, prec, p_, rec, density, 0.7); }
    return 1;
}

/* grow the region of `seed` against a private copy of the committed mask */
static void run_region(int seed, spec_t *r)
{
    memcpy(scratch, L.used, np);
    lsd_t Lx = L; Lx.used = scratch;
    int reg_size; double reg_angle;
    region_grow(&Lx, seed % L.w, seed / L.w, reg, &reg_size, &reg_angle, prec);
    r->seed = seed; r->cost = reg_size / PPR + SEED_OVH; r->big = 0; r->seg = 0;
    int ne = 0; int *ev = (int *)malloc(sizeof(int) * (reg_size + 1));
    for (int i = 0; i < reg_size; i++) ev[ne++] = reg[i].x + reg[i].y * L.w;
    if ((unsigned)reg_size >= min_reg) {
        r->big = 1;
        rect_t rec;
        region2rect(reg, reg_size, reg_angle, prec, p_, &rec);
        r->cost += 3 * (reg_size / 64.0 + 1);
        if (refine_rec(&Lx, reg, &reg_size, reg_angle, &rec, &ev, &ne, &r->cost)) {
            r->seg = 1;
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            r->s[0] = (float)(rec.x1 / 0.8); r->s[1] = (float)(rec.y1 / 0.8); r->s[2] = (float)(rec.x2 / 0.8); r->s[3] = (float)(rec.y2 / 0.8);
        }
    }
    int nf = 0; int *fn = (int *)malloc(sizeof(int) * (ne + 1));
    for (int i = 0; i < ne; i++) if (scratch[ev[i]] == USED && L.used[ev[i]] == NOTUSED) { fn[nf++] = ev[i]; scratch[ev[i]] = 3; }   /* 3: counted once */
    r->ever = ev; r->n_ever = ne; r->fin = fn; r->n_final = nf;
    r->n_blocked = g_nblocked; r->blocked = NULL;
    if (g_nblocked) { r->blocked = (int *)malloc(sizeof(int) * g_nblocked); memcpy(r->blocked, g_blocked, sizeof(int) * g_nblocked); }
    g_nblocked = 0;
    r->cost *= ROUND;
}

static int avail_px(size_t q) { return L.used[q] == NOTUSED && L.angles[q] != NOTDEF && (int)(q % L.w) < L.w - 1 && (int)(q / L.w) < L.h - 1; }

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s gray.u8 w h W [window] [val_cost]\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), W = atoi(argv[4]);
    const int WINDOW = argc > 5 ? atoi(argv[5]) : 4 * W;
    const double VAL = argc > 6 ? atof(argv[6]) : 0.5;
    if (argc > 7) TAGS_LIVE = atoi(argv[7]);
    if (argc > 8) SEE_OLDER = atoi(argv[8]);
    uint8_t *gray = (uint8_t *)malloc((size_t)w * h);
    FILE *fp = fopen(argv[1], "rb"); if (!fp || fread(gray, 1, (size_t)w * h, fp) != (size_t)w * h) { fprintf(stderr, "read failed\n"); return 1; }
    fclose(fp);
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5;
    prec = CV_PI * ANG_TH / 180; p_ = ANG_TH / 180; const double rho = QUANT / sin(prec);
    const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
    const unsigned hk = (unsigned)ceil(sigma * sqrt(2 * sprec * log(10.0)));
    const int ksize = 1 + 2 * hk;
    L.w = orc_cvround_d(w * SCALE); L.h = orc_cvround_d(h * SCALE);
    np = (size_t)L.w * L.h;
    L.scaled = (double *)malloc(sizeof(double) * np);
    double *blur = (double *)malloc(sizeof(double) * (size_t)w * h);
    gaussian_blur_f64(gray, w, h, w, blur, ksize, sigma);
    resize_linear_f64(blur, w, h, L.scaled, L.w, L.h, SCALE, SCALE);
    L.angles = (double *)malloc(sizeof(double) * np); L.modgrad = (double *)calloc(np, sizeof(double)); L.used = (uint8_t *)calloc(np, 1);
    for (int x = 0; x < L.w; x++) L.angles[(size_t)(L.h - 1) * L.w + x] = NOTDEF;
    for (int y = 0; y < L.h; y++) L.angles[(size_t)y * L.w + L.w - 1] = NOTDEF;
    for (int y = 0; y < L.h - 1; ++y)
        for (int x = 0; x < L.w - 1; ++x) {
            const int addr = y * L.w + x;
            double DA = L.scaled[addr + L.w + 1] - L.scaled[addr], BC = L.scaled[addr + 1] - L.scaled[addr + L.w];
            double gx = DA + BC, gy = DA - BC, norm = sqrt((gx * gx + gy * gy) / 4);
            L.modgrad[addr] = norm;
            L.angles[addr] = norm <= rho ? NOTDEF : orc_fast_atan2((float)gx, (float)-gy) * DEG_TO_RADS;
        }
    const double LOG_NT = 5 * (log10((double)L.w) + log10((double)L.h)) / 2 + log10(11.0);
    min_reg = (unsigned)(-LOG_NT / log10(p_));
    reg = (regpt *)malloc(sizeof(regpt) * np); scratch = (uint8_t *)malloc(np);

    /* the sequential run: segments and total cost */
    float *seq = (float *)malloc(sizeof(float) * 4 * 65536); int nseq = 0; double cost_seq = 0; int nreg_seq = 0;
    for (size_t q = 0; q < np; q++) {
        if (!avail_px(q)) continue;
        spec_t r; run_region((int)q, &r);
        for (int i = 0; i < r.n_final; i++) L.used[r.fin[i]] = USED;
        if (r.seg) { memcpy(seq + 4 * nseq, r.s, 16); nseq++; }
        cost_seq += r.cost; nreg_seq++;
        free(r.ever); free(r.fin);
    }
    memset(L.used, 0, np);

    /* the asynchronous run */
    spec_t *fl = (spec_t *)calloc(WINDOW + 1, sizeof(spec_t)); int nfl = 0;      /* in flight, ordered by seed */
    uint8_t *held = (uint8_t *)calloc(np, 1);                                     /* pixels in finished-uncommitted final sets (guess filter) */
    double *wfree = (double *)calloc(W, sizeof(double));                          /* when each worker is free */
    double now = 0; size_t frontier = 0, guess_from = 0;
    float *out = (float *)malloc(sizeof(float) * 4 * 65536); int nout = 0;
    long n_spec = 0, n_valid = 0, n_bad_seed = 0, n_conflict = 0, n_rerun = 0; double cost_wasted = 0, cost_rerun = 0, t_commit_busy = 0;
    for (;;) {
        { size_t s0 = frontier; while (s0 < np && !avail_px(s0)) s0++; if (s0 >= np) break; }
        /* hand guesses to free workers (worker 0 is the committer when it commits: model it as an extra agent, W workers speculate) */
        if (guess_from < frontier) guess_from = frontier;
        for (int k = 0; k < W && nfl < WINDOW; k++) {
            if (wfree[k] > now) continue;
            /* next guess: first available pixel at or after guess_from that no finished result holds and that is no in-flight seed */
            size_t q = guess_from;
            for (;;) {
                while (q < np && (!avail_px(q) || held[q])) q++;
                int dup = 0; for (int i = 0; i < nfl; i++) if ((size_t)fl[i].seed == q) dup = 1;
                if (!dup) break;
                q++;
            }
            if (q >= np) break;
            g_held = SEE_OLDER ? held : NULL;
            spec_t r; run_region((int)q, &r);
            g_held = NULL;
            r.t_start = now; r.t_done = now + r.cost; r.worker = k; r.state = 1; wfree[k] = r.t_done; n_spec++;
            int pos = nfl; while (pos > 0 && fl[pos - 1].seed > r.seed) { fl[pos] = fl[pos - 1]; pos--; }
            fl[pos] = r; nfl++;
            /* the region's claims are visible to the following guesses (owner tags are written as it grows; TAGS_LIVE = 0: only once it has
             * finished, and the next guess starts behind the run of available pixels this seed starts) */
            if (TAGS_LIVE) { for (int j = 0; j < r.n_ever; j++) held[r.ever[j]]++; guess_from = q + 1; }
            else { size_t e = q; while (e < np && avail_px(e) && !held[e] && (e % L.w) != (size_t)L.w - 1) e++; guess_from = e; }
        }
        /* commit whatever can be committed at `now` (the committing work is done by worker of the oldest finished result; serial) */
        int progressed = 1, acted = 0;
        while (progressed) {
            progressed = 0;
            size_t s = frontier; while (s < np && !avail_px(s)) s++;
            if (s >= np) { frontier = np; break; }
            /* drop results whose seed is before the true next seed, or no longer available */
            while (nfl && (size_t)fl[0].seed < s && fl[0].state == 1 && fl[0].t_done <= now) {
                n_bad_seed++; cost_wasted += fl[0].cost;
                for (int i = 0; i < fl[0].n_ever; i++) { if (TAGS_LIVE) held[fl[0].ever[i]]--; else held[fl[0].ever[i]] = 0; }
                free(fl[0].ever); free(fl[0].fin); memmove(fl, fl + 1, sizeof(spec_t) * --nfl);
            }
            if (nfl && (size_t)fl[0].seed < s) break;                 /* still running: wait for it to finish before it can be dropped (keeps the list ordered) */
            if (nfl && (size_t)fl[0].seed == s) {
                if (!(fl[0].state == 1 && fl[0].t_done <= now)) break;          /* the frontier's region is still growing */
                int ok = 1;
                for (int i = 0; ok && i < fl[0].n_ever; i++) if (L.used[fl[0].ever[i]] != NOTUSED) ok = 0;
                for (int i = 0; ok && i < fl[0].n_blocked; i++) if (L.used[fl[0].blocked[i]] != USED) ok = 0;
                const double vcost = VAL * (fl[0].n_ever / 64.0 + 1) * ROUND;
                now += vcost; t_commit_busy += vcost;
                if (ok) {
                    n_valid++;
                    for (int i = 0; i < fl[0].n_final; i++) { L.used[fl[0].fin[i]] = USED; }
                    for (int i = 0; i < fl[0].n_ever; i++) { if (TAGS_LIVE) held[fl[0].ever[i]]--; else held[fl[0].ever[i]] = 0; }
                    if (fl[0].seg) { memcpy(out + 4 * nout, fl[0].s, 16); nout++; }
                    frontier = s + 1;
                    free(fl[0].ever); free(fl[0].fin); memmove(fl, fl + 1, sizeof(spec_t) * --nfl);
                    progressed = 1; continue;
                }
                n_conflict++; cost_wasted += fl[0].cost;
                for (int i = 0; i < fl[0].n_ever; i++) { if (TAGS_LIVE) held[fl[0].ever[i]]--; else held[fl[0].ever[i]] = 0; }
                free(fl[0].ever); free(fl[0].fin); memmove(fl, fl + 1, sizeof(spec_t) * --nfl);
            }
            /* no usable result for seed s: grow it now against the committed mask (always valid); the committing worker does it */
            spec_t r; run_region((int)s, &r);
            n_rerun++; cost_rerun += r.cost;
            now += r.cost; t_commit_busy += r.cost;
            for (int i = 0; i < r.n_final; i++) L.used[r.fin[i]] = USED;
            if (r.seg) { memcpy(out + 4 * nout, r.s, 16); nout++; }
            frontier = s + 1; free(r.ever); free(r.fin);
            acted = 1; break;                                         /* time has passed: let the free workers take guesses first */
        }
        if (frontier >= np) break;
        if (acted) continue;
        /* results that have finished by `now` publish their final sets for the guess filter */
        /* advance time to the next event: the earliest worker completion after now */
        double nxt = 1e300;
        for (int i = 0; i < nfl; i++) if (fl[i].t_done > now && fl[i].t_done < nxt) nxt = fl[i].t_done;
        if (!TAGS_LIVE) for (int i = 0; i < nfl; i++) if (fl[i].t_done <= now) for (int j = 0; j < fl[i].n_final; j++) held[fl[i].fin[j]] = 1;
        if (nxt >= 1e300) {
            /* nothing running: either everything in flight is finished (the commit loop stalled on nothing) or no guesses left */
            if (!nfl) { size_t s = frontier; while (s < np && !avail_px(s)) s++; if (s >= np) break; guess_from = frontier; continue; }
            continue;
        }
        now = nxt;
        if (!TAGS_LIVE) for (int i = 0; i < nfl; i++) if (fl[i].t_done <= now) for (int j = 0; j < fl[i].n_final; j++) held[fl[i].fin[j]] = 1;
    }
    int same = nout == nseq && memcmp(out, seq, sizeof(float) * 4 * nseq) == 0;
    printf("W=%d window=%d: regions %d, segments %d (%s the sequential run's %d) | speculated %ld: valid %ld, seed swallowed %ld, claims met %ld; grown again at the frontier %ld\n",
           W, WINDOW, nreg_seq, nout, same ? "IDENTICAL to" : "DIFFERENT from", nseq, n_spec, n_valid, n_bad_seed, n_conflict, n_rerun);
    printf("   time (rounds): sequential %.0f, asynchronous %.0f (%.2fx) | committer busy %.0f (of which growing again %.0f), wasted speculation %.0f\n",
           cost_seq, now, cost_seq / now, t_commit_busy, cost_rerun, cost_wasted);
    return same ? 0 : 1;
}

/* lsd_spec_sim.c -- design study (not product code): how many SEQUENTIAL rounds would LSD's seed loop need if K regions were grown
 * side by side per round, each against the mask as it stands at the round's start (its own claims in a private overlay), and the longest
 * prefix committed for which
 *   (1) the j-th seed guessed at the round's start is the true next seed once the earlier regions are committed, and
 *   (2) no pixel the region ever claimed (first growth, refine's re-growth) is in the final claim of an earlier region of the round
 *       -- then every test it made had the outcome the sequential run gives (a pixel claimed by an earlier region and tested here was
 *       not aligned, so it is skipped either way).
 * The true sequence comes from the oracle run region by region; a round's guesses are made on the true mask at its start.
 * Seed guess: the first available pixel, then the starts of the following runs of available pixels in raster order (the pixels that
 * follow a seed on its row usually belong to its region), SKIP[j] runs apart.
 *
 *   gcc -O2 -o /tmp/lsd_spec_sim tools/lsd_spec_sim.c -lm && /tmp/lsd_spec_sim gray.u8 640 480 4
 */
#include <stdio.h>
#define orc_lsd_detect orc_lsd_detect_unused
#include "../oracle/lsd.c"
#undef orc_lsd_detect


typedef struct { int seed; int n_ever, n_final; int *ever, *fin; int big; long cost; } region_rec;

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s gray.u8 w h K\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), K = atoi(argv[4]);
    uint8_t *gray = (uint8_t *)malloc((size_t)w * h);
    FILE *fp = fopen(argv[1], "rb"); if (!fp || fread(gray, 1, (size_t)w * h, fp) != (size_t)w * h) { fprintf(stderr, "read failed\n"); return 1; }
    fclose(fp);
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, DENSITY_TH = 0.7;
    const double prec = CV_PI * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / sin(prec);
    const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
    const unsigned hk = (unsigned)ceil(sigma * sqrt(2 * sprec * log(10.0)));
    const int ksize = 1 + 2 * hk;
    lsd_t L;
    L.w = orc_cvround_d(w * SCALE); L.h = orc_cvround_d(h * SCALE);
    const size_t np = (size_t)L.w * L.h;
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
    const unsigned min_reg_size = (unsigned)(-LOG_NT / log10(p));
    regpt *reg = (regpt *)malloc(sizeof(regpt) * np);
    /* the true sequence, with every region's ever-claimed and final sets (from mask differences) */
    region_rec *R = (region_rec *)malloc(sizeof(region_rec) * np); int nr = 0;
    uint8_t *before = (uint8_t *)malloc(np), *ever = (uint8_t *)calloc(np, 1);
    for (int y = 0; y < L.h - 1; ++y)
        for (int x = 0; x < L.w - 1; ++x) {
            const int adx = x + y * L.w;
            if (L.used[adx] != NOTUSED || L.angles[adx] == NOTDEF) continue;
            memcpy(before, L.used, np);
            region_rec *r = &R[nr++]; r->seed = adx; r->big = 0;
            int reg_size; double reg_angle;
            region_grow(&L, x, y, reg, &reg_size, &reg_angle, prec);
            int ne = 0; int *ev = (int *)malloc(sizeof(int) * (reg_size + 1));
            for (int i = 0; i < reg_size; i++) { ev[ne++] = reg[i].x + reg[i].y * L.w; ever[ev[ne - 1]] = 1; }
            r->cost = reg_size;
            if ((unsigned)reg_size >= min_reg_size) {
                r->big = 1;
                rect_t rec;
                region2rect(reg, reg_size, reg_angle, prec, p, &rec);
                r->cost += reg_size;
                const double density = (double)reg_size / (distd(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                if (density < DENSITY_TH) r->cost += 2 * reg_size;
                refine(&L, reg, &reg_size, reg_angle, prec, p, &rec, DENSITY_TH);
                /* the re-growth may have claimed pixels outside the first region */
                for (size_t q = 0; q < np; q++) if (L.used[q] && !before[q] && !ever[q]) { ev = (int *)realloc(ev, sizeof(int) * (ne + 1)); ev[ne++] = (int)q; }
            }
            for (int i = 0; i < ne; i++) ever[ev[i]] = 0;
            r->ever = ev; r->n_ever = ne;
            int nf = 0; int *fn = (int *)malloc(sizeof(int) * (ne + 1));
            for (int i = 0; i < ne; i++) if (L.used[ev[i]] && !before[ev[i]]) fn[nf++] = ev[i];
            r->fin = fn; r->n_final = nf;
        }
    /* replay with K regions per round */
    memset(L.used, 0, np);
    for (size_t q = 0; q < np; q++) if (L.angles[q] == NOTDEF) L.used[q] = 2;       /* never available */
    for (int y = 0; y < L.h; y++) L.used[(size_t)y * L.w + L.w - 1] = 2;
    for (int x = 0; x < L.w; x++) L.used[(size_t)(L.h - 1) * L.w + x] = 2;
    uint8_t *claimed = (uint8_t *)calloc(np, 1);
    long rounds = 0, hist[65] = {0}, wrong_seed = 0, conflict = 0, cost_seq = 0, cost_par = 0;
    int t = 0;
    while (t < nr) {
        /* guesses: the first available pixel, then starts of the following runs */
        int cand[64]; int nc = 0; size_t q = R[t].seed;
        cand[nc++] = (int)q;
        while (nc < K) {
            while (q < np && L.used[q] == 0 && (q % L.w) != (size_t)L.w - 1) q++;            /* to the end of this run */
            while (q < np && L.used[q] != 0) q++;                                             /* to the start of the next */
            if (q >= np) break;
            cand[nc++] = (int)q;
        }
        int Lc = 0; long maxcost = 0;
        for (int j = 0; j < nc && t + j < nr; j++) {
            if (cand[j] != R[t + j].seed) { if (j < nc) wrong_seed++; break; }
            int ok = 1;
            for (int i = 0; ok && i < R[t + j].n_ever; i++) if (claimed[R[t + j].ever[i]]) ok = 0;
            if (!ok) { conflict++; break; }
            for (int i = 0; i < R[t + j].n_final; i++) claimed[R[t + j].fin[i]] = 1;
            if (R[t + j].cost > maxcost) maxcost = R[t + j].cost;
            Lc++;
        }
        /* a round costs its slowest region (all K grow side by side; the ones not committed are wasted) */
        for (int j = 0; j < nc && t + j < nr; j++) if (cand[j] == R[t + j].seed && R[t + j].cost > maxcost && j <= Lc) maxcost = R[t + j].cost;
        for (int j = 0; j < Lc; j++) {
            cost_seq += R[t + j].cost;
            for (int i = 0; i < R[t + j].n_final; i++) { claimed[R[t + j].fin[i]] = 0; L.used[R[t + j].fin[i]] = 1; }
        }
        cost_par += maxcost;
        rounds++; hist[Lc]++; t += Lc;
    }
    printf("K=%d regions=%d rounds=%ld regions/round=%.2f | stops: wrong seed %ld, conflict %ld | cost (points, refine-weighted): sequential %ld, side by side %ld (%.2fx)\n",
           K, nr, rounds, (double)nr / rounds, wrong_seed, conflict, cost_seq, cost_par, (double)cost_seq / cost_par);
    printf("prefix-length histogram:"); for (int i = 1; i <= K; i++) printf(" %d:%ld", i, hist[i]); printf("\n");
    return 0;
}

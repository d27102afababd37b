/* flood_gen_sim: how many GENERATIONS does floodFill's FIFO have (entries pushed by the generation before), and how large are they?
 * The generation-parallel flood kernel (csrc/peac_flood_gen.inc) processes a whole generation at once; its time is generations x round trips.
 * Build: gcc -O2 -I oracle tools/flood_gen_sim.c oracle/synth... (see tools/flood_gen_sim.py, which feeds frames through ctypes)
 * This file #includes oracle/peac.c with the statistics hook defined. */
#include <stdio.h>
static int g_ngen, g_first, g_maxgen, g_total; static long g_sum;
static int g_sizes[4096];
static void flood_stat(int gen, int k, int nq) { if (gen < 4096) g_sizes[gen] = nq - k; g_ngen = gen + 1; }
#define ORC_PEAC_FLOOD_STATS(gen, k, nq) flood_stat(gen, k, nq)
#include "../oracle/peac.c"
int flood_gen_stats(int *ngen, int *sizes, int cap) { *ngen = g_ngen; for (int i = 0; i < g_ngen && i < cap; i++) sizes[i] = g_sizes[i]; return 0; }

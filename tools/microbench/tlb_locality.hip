// tlb_locality.hip -- does the per-frame layout of an order-dependent kernel's working set matter for its load latency?
// 2048 waves (two per SIMD, as k_peac_cluster runs), four "frames" per wave (16 lanes each), every lane walks a dependent chain of
// 128-byte lines inside its frame's working set of NARR x REG bytes.  Layout A ("arrays"): NARR separate arrays indexed by frame
// (array a of frame f at a * stride_a + f * REG) -- what the plans do today.  Layout B ("slab"): one slab per frame (f * NARR * REG + a * REG).
// The same lines are touched in the same order; only the addresses differ.   hipcc -O2 --offload-arch=gfx950 tlb_locality.hip -o tlb_locality
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_fill(uint32_t *m, size_t nlines)
{
    for (size_t i = blockIdx.x * 64ull + threadIdx.x; i < nlines; i += (size_t)gridDim.x * 64) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        m[i * 32] = h;
    }
}

template <int SLAB>
__global__ __launch_bounds__(64) void k_chase(const uint32_t *__restrict__ m, int nframes, int narr, int reg_lines, int steps, unsigned long long *out, uint32_t *sink, int act, int fmod)
{
    const int lane = threadIdx.x; int f = blockIdx.x * 4 + (lane >> 4);
    if (f >= nframes || (lane & 15) >= act) return;
    f %= fmod;                                            // fmod < nframes: the same number of chains over a smaller footprint
    uint32_t v = (uint32_t)(f * 64 + lane) * 747796405u + 1u;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < steps; i++) {
        const uint32_t a = (v >> 3) % (uint32_t)narr, l = (v >> 8) % (uint32_t)reg_lines;
        const size_t line = SLAB ? ((size_t)f * narr + a) * reg_lines + l : ((size_t)a * nframes + f) * reg_lines + l;
        v = __builtin_nontemporal_load(m + line * 32) + (uint32_t)i * 40503u + (uint32_t)lane;
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) atomicAdd(out, t1 - t0);
    if (v == 0x12345u) *sink = v;
}

int main(int argc, char **argv)
{
    const int nframes = argc > 1 ? atoi(argv[1]) : 8192, narr = argc > 2 ? atoi(argv[2]) : 8, reg_kb = argc > 3 ? atoi(argv[3]) : 192, steps = argc > 4 ? atoi(argv[4]) : 4000, act = argc > 5 ? atoi(argv[5]) : 16, fmod = argc > 6 ? atoi(argv[6]) : nframes;
    const int reg_lines = reg_kb * 1024 / 128;
    const size_t nlines = (size_t)nframes * narr * reg_lines;
    uint32_t *m, *sink; unsigned long long *out;
    CK(hipMalloc(&m, nlines * 128)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&out, 8));
    k_fill<<<4096, 64>>>(m, nlines);
    CK(hipDeviceSynchronize());
    printf("# active lanes per frame %d, frames folded onto %d (%.2f GB touched)\n", act, fmod, (double)fmod * narr * reg_lines * 128 / 1e9);
    printf("# %d frames x %d arrays x %d KB = %.1f GB; %d waves of 4 frames; %d dependent steps per lane\n", nframes, narr, reg_kb, nlines * 128 / 1e9, nframes / 4, steps);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++)
        for (int slab = 0; slab < 2; slab++) {
            CK(hipMemset(out, 0, 8));
            CK(hipEventRecord(e0));
            if (slab) k_chase<1><<<nframes / 4, 64>>>(m, nframes, narr, reg_lines, steps, out, sink, act, fmod);
            else k_chase<0><<<nframes / 4, 64>>>(m, nframes, narr, reg_lines, steps, out, sink, act, fmod);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long t; CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
            printf("layout %-6s  %8.3f ms  %7.1f ns per step (launch)  %8.1f ticks per step (clock64, mean over waves)\n", slab ? "slab" : "arrays", ms, ms * 1e6 / steps, (double)t / (nframes / 4) / steps);
        }
    return 0;
}

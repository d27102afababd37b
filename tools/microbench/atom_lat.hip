// Latency of ONE wave's dependent global accesses by kind (what lsd_async.inc / the multi-worker kernels choose between):
//   hipcc -O3 --offload-arch=gfx950 atom_lat.hip -o atom_lat && ./atom_lat
// plain load (L1 may serve it), load sc0 (workgroup scope), load sc1 (agent scope), load sc0 sc1 (system), RMW atomic with return at
// workgroup / agent scope, and a store + s_waitcnt vmcnt(0) (the "release" of those kernels).  64 KB working set, lane 0 chases.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define N 16384
template <int KIND> __global__ __launch_bounds__(64) void k(unsigned *buf, int steps, unsigned *out, unsigned long long *ticks)
{
    unsigned idx = threadIdx.x * 37 % N;
    unsigned z = 0; asm volatile("" : "+v"(z));
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; s++) {
        unsigned v;
        unsigned *p = buf + idx;
        if (KIND == 0) v = *(volatile unsigned *)p;
        else if (KIND == 1) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (KIND == 2) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (KIND == 3) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (KIND == 4) v = __hip_atomic_fetch_or(p, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (KIND == 5) v = __hip_atomic_fetch_or(p, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (KIND == 6) { asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
        else if (KIND == 7) { asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
        else if (KIND == 8) { __hip_atomic_store(p, idx * 2654435761u % N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); v = idx * 40503u + 1; }
        else { __hip_atomic_store(p, idx * 2654435761u % N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); v = idx * 40503u + 1; }
        idx = v % N;
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = idx; ticks[blockIdx.x] = (unsigned long long)(t1 - t0); }
}
__global__ void k_init(unsigned *buf) { const unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < N) buf[i] = (i * 2654435761u + 12345u) % N; }
int main()
{
    unsigned *buf, *out; unsigned long long *ticks; CK(hipMalloc(&buf, N * 4)); CK(hipMalloc(&out, 1024)); CK(hipMalloc(&ticks, 8 * 256));
    const char *names[] = { "plain load (volatile)", "atomic load, workgroup scope", "atomic load, agent scope", "atomic load, system scope", "fetch_or 0 with return, workgroup scope",
                            "fetch_or 0 with return, agent scope", "global_load sc0 (asm)", "global_load nt (asm)", "store workgroup + vmcnt(0)", "store agent + vmcnt(0)" };
    const int steps = 2000;
    printf("# one wave, %d dependent accesses each (all 64 lanes chase their own chain), 64 KB set; ns per access; and with 16 such waves on 16 CUs at once\n", steps);
    for (int kind = 0; kind < 10; kind++) {
        for (int nb : { 1, 16 }) {
            hipLaunchKernelGGL(k_init, dim3(N / 256), dim3(256), 0, 0, buf); CK(hipDeviceSynchronize());
            unsigned long long h[256];
#define RUN(K) case K: hipLaunchKernelGGL(k<K>, dim3(nb), dim3(64), 0, 0, buf, steps, out, ticks); break;
            for (int rep = 0; rep < 2; rep++) { switch (kind) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) } CK(hipDeviceSynchronize()); }
            CK(hipMemcpy(h, ticks, 8 * nb, hipMemcpyDeviceToHost));
            double mx = 0; for (int i = 0; i < nb; i++) if (h[i] > mx) mx = (double)h[i];
            printf("%-44s %2d wave(s): %7.1f ns\n", names[kind], nb, mx * 10.0 / steps);
        }
    }
    return 0;
}

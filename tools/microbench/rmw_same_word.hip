// How long a wave waits (s_waitcnt vmcnt(0)) after k of its lanes issue ONE global update each, without return:
//   atomic_or to k bits of ONE word | atomic_or to k different words | plain byte stores to k bytes of one word | k different lines.
// What lsd_async.inc found the hard way (round 4): a region's pixels are neighbours, their bits share mask words, and read-modify-writes
// to one word are performed one after the other in the L2.      hipcc -O3 --offload-arch=gfx950 rmw_same_word.hip -o rmw_same_word
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int KIND> __global__ __launch_bounds__(64) void k(unsigned *buf, int k_lanes, int steps, unsigned long long *ticks)
{
    const int lane = threadIdx.x;
    unsigned char *bytes = reinterpret_cast<unsigned char *>(buf);
    long long total = 0;
    for (int s = 0; s < steps; s++) {
        const unsigned base = (unsigned)(s * 97 % 4096) * 64u;           // another place each step (words; 1 MB set, L2-resident)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t0 = wall_clock64();
        if (lane < k_lanes) {
            if (KIND == 0) __hip_atomic_fetch_or(&buf[base], 1u << (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 1) __hip_atomic_fetch_or(&buf[base + lane], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 2) __hip_atomic_store(&bytes[4 * base + lane], (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_or(&buf[base + 32 * lane], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        total += wall_clock64() - t0;
    }
    if (lane == 0) ticks[blockIdx.x] = (unsigned long long)total;
}
int main()
{
    unsigned *buf; unsigned long long *ticks; CK(hipMalloc(&buf, 4096 * 64 * 4 + 65536)); CK(hipMalloc(&ticks, 64)); CK(hipMemset(buf, 0, 4096 * 64 * 4 + 65536));
    const char *names[] = { "atomic_or, k bits of ONE word", "atomic_or, k adjacent words (one line)", "byte stores, k bytes of one line", "atomic_or, k words in k lines" };
    const int steps = 2000;
    printf("# one wave; ns from the issue of k lanes' updates (no return value) to the end of s_waitcnt vmcnt(0); %d repetitions\n", steps);
    printf("%-42s", "k ="); for (int kl : { 1, 2, 4, 8, 16, 32 }) printf(" %7d", kl); printf("\n");
    for (int kind = 0; kind < 4; kind++) {
        printf("%-42s", names[kind]);
        for (int kl : { 1, 2, 4, 8, 16, 32 }) {
            unsigned long long h = 0;
#define RUN(K) case K: hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, buf, kl, steps, ticks); break;
            for (int rep = 0; rep < 2; rep++) { switch (kind) { RUN(0) RUN(1) RUN(2) RUN(3) } CK(hipDeviceSynchronize()); }
            CK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
            printf(" %7.0f", (double)h * 10.0 / steps);
        }
        printf("\n");
    }
    return 0;
}

// Random dependent 8-byte loads over a working set of S bytes: latency per load vs S (address-translation reach).
//   hipcc -O3 --offload-arch=gfx950 tlb_chase.hip -o tlb_chase && ./tlb_chase
// Every lane follows its own chain (64 scattered lines per wave-level load, like the AHC / flood / LSD kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

__global__ void k_init(uint64_t *buf, uint64_t nlines)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nlines; i += (uint64_t)gridDim.x * blockDim.x)
        buf[i * 16] = mix(i * 0x9E3779B97F4A7C15ULL + 12345) % nlines;          // next line index, one per 128-byte line
}

__global__ __launch_bounds__(64) void k_chase(const uint64_t *buf, uint64_t nlines, int steps, uint64_t *sink)
{
    uint64_t idx = mix(blockIdx.x * 64ull + threadIdx.x + 777) % nlines;
    for (int s = 0; s < steps; s++) idx = buf[idx * 16];
    if (idx == 0xFFFFFFFFFFFFFFFFull) sink[0] = idx;
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 8192, steps = 200;
    size_t free_b, tot_b; CK(hipMemGetInfo(&free_b, &tot_b));
    uint64_t *sink; CK(hipMalloc(&sink, 8));
    const double sizes_gb[] = { 0.25, 1, 4, 16, 64, 128, 200 };
    size_t maxb = (size_t)(200.0 * (1ull << 30));
    if (maxb > free_b - (4ull << 30)) maxb = free_b - (4ull << 30);
    uint64_t *buf; CK(hipMalloc(&buf, maxb));
    printf("waves %d (x64 chains), %d dependent loads each; free %.0f GB\n", waves, steps, free_b / 1e9);
    for (double g : sizes_gb) {
        size_t bytes = (size_t)(g * (1ull << 30)); if (bytes > maxb) bytes = maxb;
        const uint64_t nlines = bytes / 128;
        hipLaunchKernelGGL(k_init, dim3(65536), dim3(256), 0, 0, buf, nlines);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, 0, buf, nlines, 20, sink);      // warm
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chase, dim3(waves), dim3(64), 0, 0, buf, nlines, steps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)waves * 64 * steps;
        printf("working set %7.2f GB: %8.2f ms, %7.1f ns per dependent step, %6.2f G loads/s (%.0f GB/s of 128-byte lines)\n",
               bytes / 1073741824.0, ms, ms * 1e6 / steps, loads / ms / 1e6, loads * 128 / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}

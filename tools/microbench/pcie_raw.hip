// pcie_raw.hip -- what the host link delivers, measured raw: one 1 GiB pinned buffer per direction,
// hipMemcpyAsync host -> device, device -> host and both at once on two streams, with the pinned pages bound to each NUMA
// node in turn (mbind before the first touch, then hipHostRegister), plus a hipHostMalloc buffer (the runtime's own placement).
//
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/pcie_raw.hip -o tools/microbench/pcie_raw && tools/microbench/pcie_raw > profiles/rNN_pcie_raw.txt
//
// The library's upload leg (hvo_batch_upload) is compared against the best figure printed here.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static long mbind_node(void *p, size_t bytes, int node)
{
    unsigned long mask[16] = { 0 };
    mask[node / 64] |= 1ul << (node % 64);
    return syscall(SYS_mbind, p, bytes, 2 /* MPOL_BIND */, mask, 1024ul, 0u);
}

struct Rates { double h2d, d2h, bi_h2d, bi_d2h; };

static Rates measure(void *h_in, void *h_out, void *d_a, void *d_b, size_t bytes, hipStream_t s0, hipStream_t s1, int reps)
{
    Rates r;
    CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s1));
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    r.h2d = reps * (double)bytes / (now() - t0) / 1e9;
    t0 = now();
    for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    r.d2h = reps * (double)bytes / (now() - t0) / 1e9;
    t0 = now();
    for (int i = 0; i < reps; i++) { CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s1)); }
    CK(hipStreamSynchronize(s0)); const double ta = now() - t0;
    CK(hipStreamSynchronize(s1)); const double tb = now() - t0;
    r.bi_h2d = reps * (double)bytes / ta / 1e9; r.bi_d2h = reps * (double)bytes / tb / 1e9;
    return r;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 1024) << 20;
    const int reps = 5;
    CK(hipSetDevice(0));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("raw host link, %s, buffer %zu MiB per direction, %d copies per figure (GB/s = 1e9 B/s)\n", pr.name, bytes >> 20, reps);
    void *d_a, *d_b; CK(hipMalloc(&d_a, bytes)); CK(hipMalloc(&d_b, bytes));
    CK(hipMemset(d_a, 1, bytes)); CK(hipMemset(d_b, 2, bytes));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    {
        void *hi, *ho; CK(hipHostMalloc(&hi, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&ho, bytes, hipHostMallocDefault));
        memset(hi, 3, bytes); memset(ho, 4, bytes);
        const Rates r = measure(hi, ho, d_a, d_b, bytes, s0, s1, reps);
        printf("  hipHostMalloc (runtime placement)   H2D %6.1f  D2H %6.1f  both at once: H2D %6.1f + D2H %6.1f = %6.1f\n", r.h2d, r.d2h, r.bi_h2d, r.bi_d2h, r.bi_h2d + r.bi_d2h);
        CK(hipHostFree(hi)); CK(hipHostFree(ho));
    }
    std::vector<int> nodes;
    if (DIR *d = opendir("/sys/devices/system/node")) {
        while (dirent *e = readdir(d)) if (!strncmp(e->d_name, "node", 4) && e->d_name[4] >= '0' && e->d_name[4] <= '9') nodes.push_back(atoi(e->d_name + 4));
        closedir(d);
    }
    for (int node : nodes) {
        void *hi = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        void *ho = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (hi == MAP_FAILED || ho == MAP_FAILED) { printf("  node %d: mmap failed\n", node); continue; }
        const long b0 = mbind_node(hi, bytes, node), b1 = mbind_node(ho, bytes, node);
        if (b0 || b1) { printf("  node %d: mbind refused (not allowed for this user or node without memory)\n", node); munmap(hi, bytes); munmap(ho, bytes); continue; }
        memset(hi, 3, bytes); memset(ho, 4, bytes);
        if (hipHostRegister(hi, bytes, hipHostRegisterDefault) != hipSuccess || hipHostRegister(ho, bytes, hipHostRegisterDefault) != hipSuccess) {
            printf("  node %d: hipHostRegister failed\n", node); (void)hipGetLastError(); munmap(hi, bytes); munmap(ho, bytes); continue;
        }
        const Rates r = measure(hi, ho, d_a, d_b, bytes, s0, s1, reps);
        printf("  pinned pages bound to NUMA node %-3d  H2D %6.1f  D2H %6.1f  both at once: H2D %6.1f + D2H %6.1f = %6.1f\n", node, r.h2d, r.d2h, r.bi_h2d, r.bi_d2h, r.bi_h2d + r.bi_d2h);
        CK(hipHostUnregister(hi)); CK(hipHostUnregister(ho)); munmap(hi, bytes); munmap(ho, bytes);
    }
    // the same bytes in 2048 frame-sized pieces (307 200 B grey images), one call each and as one 2-D copy: what the call count costs
    {
        void *hi; CK(hipHostMalloc(&hi, bytes, hipHostMallocDefault)); memset(hi, 5, bytes);
        const size_t fb = 640 * 480; const int nf = (int)(bytes / fb) < 2048 ? (int)(bytes / fb) : 2048;
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int f = 0; f < nf; f++) CK(hipMemcpyAsync((char *)d_a + f * fb, (char *)hi + f * fb, fb, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        const double per_call = nf * (double)fb / (now() - t0) / 1e9;
        t0 = now();
        CK(hipMemcpy2DAsync(d_a, fb + 256, hi, fb, fb, nf, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        const double two_d = nf * (double)fb / (now() - t0) / 1e9;
        t0 = now();
        CK(hipMemcpyAsync(d_a, hi, nf * fb, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        const double one = nf * (double)fb / (now() - t0) / 1e9;
        printf("  %d frames of %zu B host -> device: one call per frame %6.1f, one strided 2-D copy (device pitch %zu) %6.1f, one contiguous copy %6.1f\n", nf, fb, per_call, fb + 256, two_d, one);
        CK(hipHostFree(hi));
    }
    return 0;
}

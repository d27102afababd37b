// cos / sin of a float angle evaluated in double on the device (ocml) and on the host (glibc), rounded to float: the LBD band directions
// (binary_descriptor_custom.cpp:1121-1139 as oracle/lbd.c restates it) start from these.   hipcc --offload-arch=gfx950 -O2 cos_check.hip -o cos_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const float *a, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = cos((double)a[i]); out[2 * i + 1] = sin((double)a[i]); }
}
int main()
{
    const int n = 1 << 20;
    float *h = new float[n]; double *o = new double[2 * n];
    h[0] = -1.5707964f; h[1] = 1.5707964f; h[2] = 3.1415927f; h[3] = -3.1415927f; h[4] = 0.0f; h[5] = 0.78539816f;
    unsigned s = 12345u;
    for (int i = 6; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 6.2831853f; }
    // the angles atan2 of small integer differences gives (key lines between pixel centres)
    for (int i = 6, y = -20; y <= 20; y++) for (int x = -20; x <= 20; x++, i++) h[i] = (float)atan2((double)((float)y * 0.125f), (double)((float)x * 0.125f));
    float *da; double *dd;
    hipMalloc(&da, n * 4); hipMalloc(&dd, n * 16); hipMemcpy(da, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, n, dd);
    hipMemcpy(o, dd, n * 16, hipMemcpyDeviceToHost);
    long dd_diff = 0, f_diff = 0;
    for (int i = 0; i < n; i++) {
        const double hc = cos((double)h[i]), hs = sin((double)h[i]);
        const bool d1 = memcmp(&hc, &o[2 * i], 8) != 0, d2 = memcmp(&hs, &o[2 * i + 1], 8) != 0;
        dd_diff += d1 + d2;
        const bool f1 = (float)hc != (float)o[2 * i], f2 = (float)hs != (float)o[2 * i + 1];
        if ((f1 || f2) && f_diff < 8) printf("angle %.9g: cos host %.17g device %.17g | sin host %.17g device %.17g\n", h[i], hc, o[2 * i], hs, o[2 * i + 1]);
        f_diff += f1 + f2;
    }
    printf("angle %.9g: cos host %.17g device %.17g\n", h[0], cos((double)h[0]), o[0]);
    printf("%d angles: %ld double results differ in the last place, %ld differ after rounding to float\n", n, dd_diff, f_diff);
    return 0;
}

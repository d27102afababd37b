// float sqrt and division on the device against the host's correctly rounded ones: `__fsqrt_rn` of this ROCm's headers is
// __ocml_native_sqrt_f32 (v_sqrt_f32, 1 ulp) unless OCML_BASIC_ROUNDED_OPERATIONS is defined -- found by a 1-bit LBD mismatch in a long soak;
// sqrtf() and `/` are correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off sqrt_check.hip -o sqrt_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const float *a, int n, float *o)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { o[4 * i] = __fsqrt_rn(a[i]); o[4 * i + 1] = sqrtf(a[i]); o[4 * i + 2] = __fdiv_rn(1.f, a[i]); o[4 * i + 3] = 1.f / sqrtf(a[i]); }
}
int main()
{
    const int n = 1 << 22;
    float *h = new float[n], *o = new float[4 * n];
    unsigned s = 777u;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; const unsigned bits = (s >> 9) | ((100u + (s & 63u)) << 23); memcpy(&h[i], &bits, 4); }
    h[0] = 0x1.b8a4d0p-5f;
    float *da, *dd;
    hipMalloc(&da, n * 4); hipMalloc(&dd, n * 16); hipMemcpy(da, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, n, dd);
    hipMemcpy(o, dd, n * 16, hipMemcpyDeviceToHost);
    long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < n; i++) {
        const float r = sqrtf(h[i]);
        d0 += o[4 * i] != r; d1 += o[4 * i + 1] != r; d2 += o[4 * i + 2] != 1.f / h[i]; d3 += o[4 * i + 3] != 1.f / r;
    }
    printf("%d values: __fsqrt_rn differs from the host's sqrtf on %ld, sqrtf on %ld, __fdiv_rn(1, x) on %ld, 1 / sqrtf on %ld\n", n, d0, d1, d2, d3);
    printf("sqrt(0x1.b8a4d0p-5): host %a, __fsqrt_rn %a, sqrtf %a\n", sqrtf(h[0]), o[0], o[1]);
    return 0;
}

// Which lane holds which element of v_mfma_i32_16x16x64_i8's operands on gfx950?  (cdna_hip_programming.md gives the bf16 maps and says
// "other dtypes: check the map with exact integer data".)  A[16][64], B[64][16] with asymmetric integer data; two hypotheses for the k index
// of byte j (0..15) of lane group g = lane >> 4:   H0: k = 16 g + j      H1: k = 8 g + j (j < 8), 32 + 8 g + (j - 8) (j >= 8)
// and the C map col = lane & 15, row = 4 (lane >> 4) + reg.  (Both hypotheses give the right C: A and B are permuted alike, and a sum over k does not
// care -- what a kernel needs is only that byte j of lane group g of A meets byte j of lane group g of B, and the C map.)  Also times a stream
// of independent MFMAs (cycles per instruction) beside a stream of v_dot4.
//   hipcc -O3 --offload-arch=gfx950 mfma_i8_layout.hip -o mfma_i8_layout && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_one(const v4i *a, const v4i *b, v4i *c)
{
    const int l = threadIdx.x;
    v4i acc = { 0, 0, 0, 0 };
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[l], b[l], acc, 0, 0, 0);
    c[l] = acc;
}

template <int MODE> __global__ __launch_bounds__(256) void k_rate(const v4i *a, const v4i *b, v4i *c, int iters, unsigned *cyc)
{
    const int l = threadIdx.x & 63;
    v4i A = a[l], B = b[l];
    v4i acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v4i){ i, 0, 0, 0 };
    unsigned d[8] = { 1, 2, 3, 4, 5, 6, 7, 8 };
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE != 1) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[i], 0, 0, 0);
            if (MODE != 0) { d[i] = __builtin_amdgcn_udot4(A.x, d[i], d[i], false); d[i] = __builtin_amdgcn_udot4(A.y, d[i], d[i], false); d[i] = __builtin_amdgcn_udot4(A.z, d[i], d[i], false); d[i] = __builtin_amdgcn_udot4(A.w, d[i], d[i], false); }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    v4i s = acc[0]; for (int i = 1; i < 8; i++) s += acc[i];
    unsigned ds = 0; for (int i = 0; i < 8; i++) ds += d[i];
    s.x += (int)ds;
    c[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = (unsigned)(t1 - t0);
}

int main()
{
    int8_t A[16][64], B[64][16];
    for (int m = 0; m < 16; m++) for (int k = 0; k < 64; k++) A[m][k] = (int8_t)(((m * 7 + k * 3 + (m * k) % 5) % 23) - 11);
    for (int k = 0; k < 64; k++) for (int n = 0; n < 16; n++) B[k][n] = (int8_t)(((k * 5 + n * 11 + (k ^ n)) % 19) - 9);
    int ref[16][16];
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { int s = 0; for (int k = 0; k < 64; k++) s += (int)A[m][k] * (int)B[k][n]; ref[m][n] = s; }
    v4i *da, *db, *dc; unsigned *dcy;
    CK(hipMalloc(&da, 64 * 16)); CK(hipMalloc(&db, 64 * 16)); CK(hipMalloc(&dc, (size_t)256 * 8 * 256 * 16 + 4096));      // k_rate: 2048 workgroups x 256 lanes x 16 bytes
    CK(hipMalloc(&dcy, 4));
    for (int hyp = 0; hyp < 2; hyp++) {
        int8_t ha[64][16], hb[64][16];
        for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            const int g = l >> 4, k = hyp == 0 ? 16 * g + j : (j < 8 ? 8 * g + j : 32 + 8 * g + (j - 8));
            ha[l][j] = A[l & 15][k]; hb[l][j] = B[k][l & 15];
        }
        CK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dc);
        int hc[64][4]; CK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
        int bad = 0, badT = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
            if (hc[l][r] != ref[4 * (l >> 4) + r][l & 15]) bad++;
            if (hc[l][r] != ref[l & 15][4 * (l >> 4) + r]) badT++;
        }
        printf("hypothesis H%d: %d mismatches with C[row = 4 (lane >> 4) + reg][col = lane & 15], %d with the transposed map\n", hyp, bad, badT);
    }
    for (int mode = 0; mode < 3; mode++) {
        const int iters = 2000, nblk = 256 * 8;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(nblk), dim3(256), 0, 0, da, db, dc, iters, dcy);
            else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(nblk), dim3(256), 0, 0, da, db, dc, iters, dcy);
            else hipLaunchKernelGGL(k_rate<2>, dim3(nblk), dim3(256), 0, 0, da, db, dc, iters, dcy);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned cy; CK(hipMemcpy(&cy, dcy, 4, hipMemcpyDeviceToHost));
        // 8 workgroups of 4 waves per CU = 8 waves per SIMD; per wave 8 * iters MFMAs (and / or 32 * iters dot4)
        const double per_simd = (double)ms * 1e6 / (8.0 * iters * 8.0);
        printf("%s: %.3f ms, %.2f ns per loop body slot per SIMD (8 waves per SIMD; body = %s), wave 0: %.1f cycles per body\n",
               mode == 0 ? "mfma only" : mode == 1 ? "dot4 only (4 per body)" : "mfma + 4 dot4", ms, per_simd,
               mode == 0 ? "1 mfma" : mode == 1 ? "4 dot4" : "1 mfma + 4 dot4", (double)cy / (8.0 * iters));
    }
    return 0;
}

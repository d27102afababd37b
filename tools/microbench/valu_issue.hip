// How fast does a gfx950 SIMD issue the vector instructions this front-end is made of?
//   hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue && ./valu_issue > profiles/rNN_valu_issue.txt
// For every instruction: a stream of 64 INDEPENDENT copies per loop iteration (16 accumulators x 4), W waves per SIMD for
// W = 1, 2, 4, 8 (one workgroup of 4 W waves per CU, 256 workgroups: every CU's four SIMDs hold W waves each; the HW_ID
// histogram of workgroup 0 is printed so that the placement is seen, not assumed).  A wave measures itself with s_memtime
// (shader cycles) and the launch is timed with hipEvents.  Reported: cycles and ns per wave-instruction PER SIMD
// = time / (instructions per wave x W) -- the ns figure is what bench.py's `valu_issue_frac` multiplies the SQ counters'
// instruction counts by -- and the same for ONE dependent chain (16 copies on one accumulator: what a lone wave sees).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define D16(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)

// every form: accumulator %i (read and written), two loop-invariant sources %16, %17
#define ACC16 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])

// one kernel per instruction; T = accumulator type (uint32_t, float, double)
#define DEF_KERNEL(NAME, T, INS)                                                                                         \
    template <bool DEP> __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, uint32_t *hwid, int iters, T s0, T s1) \
    {                                                                                                                    \
        T a[16];                                                                                                         \
        for (int i = 0; i < 16; i++) a[i] = (T)(threadIdx.x * 3 + i + 1);                                                \
        __syncthreads();                                                                                                 \
        const uint64_t t0 = __builtin_readcyclecounter();                                                                \
        for (int it = 0; it < iters; it++) {                                                                             \
            if (DEP) { asm volatile(D16(INS) D16(INS) D16(INS) D16(INS) : ACC16 : "v"(s0), "v"(s1) : "vcc", "s20", "s21", "scc"); }                   \
            else     { asm volatile(R16(INS) R16(INS) R16(INS) R16(INS) : ACC16 : "v"(s0), "v"(s1) : "vcc", "s20", "s21", "scc"); }                   \
        }                                                                                                                \
        const uint64_t t1 = __builtin_readcyclecounter();                                                                \
        T s = a[0]; for (int i = 1; i < 16; i++) s += a[i];                                                              \
        if ((threadIdx.x & 63) == 0) {                                                                                   \
            const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                           \
            out[w] = (uint32_t)(t1 - t0);                                                                                \
            uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));                                \
            hwid[w] = id;                                                                                                \
        }                                                                                                                \
        if (s == (T)0x7fffffff && iters < 0) out[0] = 1;                                                                 \
    }

#define I_ADD(i)      "v_add_u32 %" #i ", %" #i ", %16\n"
#define I_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %16, %17\n"
#define I_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 1, %16\n"
#define I_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
#define I_MIN3(i)     "v_min3_i32 %" #i ", %" #i ", %16, %17\n"
#define I_MAX3U(i)    "v_max3_u32 %" #i ", %" #i ", %16, %17\n"
#define I_MINU(i)     "v_min_u32 %" #i ", %" #i ", %16\n"
#define I_DOT4(i)     "v_dot4_u32_u8 %" #i ", %16, %17, %" #i "\n"
#define I_DOT2(i)     "v_dot2_u32_u16 %" #i ", %16, %17, %" #i "\n"
#define I_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %16, %17\n"
#define I_ALIGNB(i)   "v_alignbyte_b32 %" #i ", %" #i ", %16, 1\n"
#define I_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 3\n"
#define I_BFE(i)      "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define I_SDWA(i)     "v_add_u32_sdwa %" #i ", %" #i ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define I_SDWASUB(i)  "v_sub_u16_sdwa %" #i ", %" #i ", %16 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_3\n"
#define I_PKMINU16(i) "v_pk_min_u16 %" #i ", %" #i ", %16\n"
#define I_PKMAXU16(i) "v_pk_max_u16 %" #i ", %" #i ", %16\n"
#define I_PKSUBU16(i) "v_pk_sub_u16 %" #i ", %" #i ", %16\n"
#define I_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %16\n"
#define I_PKMADU16(i) "v_pk_mad_u16 %" #i ", %" #i ", %16, %17\n"
#define I_PKMULLO(i)  "v_pk_mul_lo_u16 %" #i ", %" #i ", %16\n"
#define I_MULU24(i)   "v_mul_u32_u24 %" #i ", %" #i ", %16\n"
#define I_MADU24(i)   "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n"
#define I_MULLO(i)    "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define I_MADU64(i)   "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n"
#define I_SAD(i)      "v_sad_u8 %" #i ", %" #i ", %16, %17\n"
#define I_MSAD(i)     "v_msad_u8 %" #i ", %" #i ", %16, %17\n"
#define I_CNDMASK(i)  "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define I_CMPCND(i)   "v_cmp_lt_u32 vcc, %" #i ", %16\n v_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define I_BCNT(i)     "v_bcnt_u32_b32 %" #i ", %16, %" #i "\n"
#define I_MBCNT(i)    "v_mbcnt_lo_u32_b32 %" #i ", %16, %" #i "\n"
#define I_MOVDPP(i)   "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(i)   "v_add_u32_dpp %" #i ", %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_XOR(i)      "v_xor_b32 %" #i ", %" #i ", %16\n"
#define I_LSHR(i)     "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define I_FMAF32(i)   "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define I_ADDF32(i)   "v_add_f32 %" #i ", %" #i ", %16\n"
#define I_PKFMAF32(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define I_MULF64(i)   "v_mul_f64 %" #i ", %" #i ", %16\n"
#define I_ADDF64(i)   "v_add_f64 %" #i ", %" #i ", %16\n"
#define I_FMAF64(i)   "v_fma_f64 %" #i ", %" #i ", %16, %17\n"
#define I_CVTU8(i)    "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define I_RCPF32(i)   "v_rcp_f32 %" #i ", %" #i "\n"
#define I_SQRTF64(i)  "v_sqrt_f64 %" #i ", %" #i "\n"
#define I_RCPF64(i)   "v_rcp_f64 %" #i ", %" #i "\n"
#define I_READLANE(i) "v_readlane_b32 s20, %" #i ", 3\n"
#define I_LDSRD(i)    "ds_read_b32 %" #i ", %16\n"
// round 5: do SCALAR instructions take issue slots from the vector ones?  (the step's time matches (VALU + SALU) x 1.77 ns, not VALU alone)
#define I_SADD(i)        "s_add_u32 s20, s20, 1\n"
#define I_XOR_SADD(i)    "v_xor_b32 %" #i ", %" #i ", %16\n s_add_u32 s20, s20, 1\n"
#define I_DOT4_SADD(i)   "v_dot4_u32_u8 %" #i ", %16, %17, %" #i "\n s_add_u32 s20, s20, 1\n"
#define I_DOT4_SADD2(i)  "v_dot4_u32_u8 %" #i ", %16, %17, %" #i "\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 7\n"
#define I_FMA64_SADD(i)  "v_fma_f64 %" #i ", %" #i ", %16, %17\n s_add_u32 s20, s20, 1\n"
#define I_DOT4_NOP(i)    "v_dot4_u32_u8 %" #i ", %16, %17, %" #i "\n s_nop 0\n"

DEF_KERNEL(add_u32, uint32_t, I_ADD)
DEF_KERNEL(add3_u32, uint32_t, I_ADD3)
DEF_KERNEL(lshl_add_u32, uint32_t, I_LSHLADD)
DEF_KERNEL(and_or_b32, uint32_t, I_ANDOR)
DEF_KERNEL(xor_b32, uint32_t, I_XOR)
DEF_KERNEL(lshrrev_b32, uint32_t, I_LSHR)
DEF_KERNEL(min_u32, uint32_t, I_MINU)
DEF_KERNEL(min3_i32, uint32_t, I_MIN3)
DEF_KERNEL(max3_u32, uint32_t, I_MAX3U)
DEF_KERNEL(dot4_u32_u8, uint32_t, I_DOT4)
DEF_KERNEL(dot2_u32_u16, uint32_t, I_DOT2)
DEF_KERNEL(perm_b32, uint32_t, I_PERM)
DEF_KERNEL(alignbyte_b32, uint32_t, I_ALIGNB)
DEF_KERNEL(alignbit_b32, uint32_t, I_ALIGNBIT)
DEF_KERNEL(bfe_u32, uint32_t, I_BFE)
DEF_KERNEL(add_u32_sdwa, uint32_t, I_SDWA)
DEF_KERNEL(sub_u16_sdwa, uint32_t, I_SDWASUB)
DEF_KERNEL(pk_min_u16, uint32_t, I_PKMINU16)
DEF_KERNEL(pk_max_u16, uint32_t, I_PKMAXU16)
DEF_KERNEL(pk_sub_u16, uint32_t, I_PKSUBU16)
DEF_KERNEL(pk_add_u16, uint32_t, I_PKADDU16)
DEF_KERNEL(pk_mad_u16, uint32_t, I_PKMADU16)
DEF_KERNEL(pk_mul_lo_u16, uint32_t, I_PKMULLO)
DEF_KERNEL(mul_u32_u24, uint32_t, I_MULU24)
DEF_KERNEL(mad_u32_u24, uint32_t, I_MADU24)
DEF_KERNEL(mul_lo_u32, uint32_t, I_MULLO)
DEF_KERNEL(sad_u8, uint32_t, I_SAD)
DEF_KERNEL(msad_u8, uint32_t, I_MSAD)
DEF_KERNEL(cndmask_b32, uint32_t, I_CNDMASK)
DEF_KERNEL(cmp_cndmask, uint32_t, I_CMPCND)
DEF_KERNEL(bcnt_u32, uint32_t, I_BCNT)
DEF_KERNEL(mbcnt_lo, uint32_t, I_MBCNT)
DEF_KERNEL(mov_dpp, uint32_t, I_MOVDPP)
DEF_KERNEL(add_u32_dpp, uint32_t, I_ADDDPP)
DEF_KERNEL(cvt_f32_ubyte1, uint32_t, I_CVTU8)
DEF_KERNEL(fma_f32, float, I_FMAF32)
DEF_KERNEL(add_f32, float, I_ADDF32)
DEF_KERNEL(rcp_f32, float, I_RCPF32)
DEF_KERNEL(pk_fma_f32, double, I_PKFMAF32)
DEF_KERNEL(mul_f64, double, I_MULF64)
DEF_KERNEL(add_f64, double, I_ADDF64)
DEF_KERNEL(fma_f64, double, I_FMAF64)
DEF_KERNEL(sqrt_f64, double, I_SQRTF64)
DEF_KERNEL(rcp_f64, double, I_RCPF64)
DEF_KERNEL(s_add_only, uint32_t, I_SADD)
DEF_KERNEL(xor_plus_sadd, uint32_t, I_XOR_SADD)
DEF_KERNEL(dot4_plus_sadd, uint32_t, I_DOT4_SADD)
DEF_KERNEL(dot4_plus_2salu, uint32_t, I_DOT4_SADD2)
DEF_KERNEL(fma64_plus_sadd, double, I_FMA64_SADD)
DEF_KERNEL(dot4_plus_snop, uint32_t, I_DOT4_NOP)

struct Entry { const char *name; int per_copy; void (*run)(int, bool, int, uint32_t *, uint32_t *); };

template <typename T, typename K> static void launch(K kern, int wps, int iters, uint32_t *out, uint32_t *hwid, int ncu)
{
    // wps <= 4: one workgroup of 4*wps waves per CU; wps == 8: two workgroups of 16 waves per CU
    const int wg_waves = wps <= 4 ? 4 * wps : 16;
    const int nblk = ncu * (wps <= 4 ? 1 : wps / 4);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(wg_waves * 64), 0, 0, out, hwid, iters, (T)3, (T)5);
}

#define ENTRY(NAME, T, PER) { #NAME, PER, [](int wps, bool dep, int iters, uint32_t *out, uint32_t *hwid) {             \
        int dev, ncu; hipGetDevice(&dev); hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);       \
        if (dep) launch<T>(k_##NAME<true>, wps, iters, out, hwid, ncu); else launch<T>(k_##NAME<false>, wps, iters, out, hwid, ncu); } }

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int dev = 0, ncu = 0, clk = 0; CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int maxw = ncu * 32 * 2;
    uint32_t *out, *hwid; CK(hipMalloc(&out, maxw * 4)); CK(hipMalloc(&hwid, maxw * 4));
    std::vector<uint32_t> h(maxw), hid(maxw);
    CK(hipMemset(out, 0, maxw * 4));
    const Entry tab[] = {
        ENTRY(add_u32, uint32_t, 1), ENTRY(add3_u32, uint32_t, 1), ENTRY(lshl_add_u32, uint32_t, 1), ENTRY(and_or_b32, uint32_t, 1),
        ENTRY(xor_b32, uint32_t, 1), ENTRY(lshrrev_b32, uint32_t, 1), ENTRY(min_u32, uint32_t, 1), ENTRY(min3_i32, uint32_t, 1),
        ENTRY(max3_u32, uint32_t, 1), ENTRY(dot4_u32_u8, uint32_t, 1), ENTRY(dot2_u32_u16, uint32_t, 1), ENTRY(perm_b32, uint32_t, 1),
        ENTRY(alignbyte_b32, uint32_t, 1), ENTRY(alignbit_b32, uint32_t, 1), ENTRY(bfe_u32, uint32_t, 1), ENTRY(add_u32_sdwa, uint32_t, 1),
        ENTRY(sub_u16_sdwa, uint32_t, 1), ENTRY(pk_min_u16, uint32_t, 1), ENTRY(pk_max_u16, uint32_t, 1), ENTRY(pk_sub_u16, uint32_t, 1),
        ENTRY(pk_add_u16, uint32_t, 1), ENTRY(pk_mad_u16, uint32_t, 1), ENTRY(pk_mul_lo_u16, uint32_t, 1), ENTRY(mul_u32_u24, uint32_t, 1),
        ENTRY(mad_u32_u24, uint32_t, 1), ENTRY(mul_lo_u32, uint32_t, 1), ENTRY(sad_u8, uint32_t, 1), ENTRY(msad_u8, uint32_t, 1),
        ENTRY(cndmask_b32, uint32_t, 1), ENTRY(cmp_cndmask, uint32_t, 2), ENTRY(bcnt_u32, uint32_t, 1), ENTRY(mbcnt_lo, uint32_t, 1),
        ENTRY(mov_dpp, uint32_t, 1), ENTRY(add_u32_dpp, uint32_t, 1), ENTRY(cvt_f32_ubyte1, uint32_t, 1),
        ENTRY(fma_f32, float, 1), ENTRY(add_f32, float, 1), ENTRY(rcp_f32, float, 1), ENTRY(pk_fma_f32, double, 1),
        ENTRY(mul_f64, double, 1), ENTRY(add_f64, double, 1), ENTRY(fma_f64, double, 1), ENTRY(sqrt_f64, double, 1), ENTRY(rcp_f64, double, 1),
        ENTRY(s_add_only, uint32_t, 1), ENTRY(xor_plus_sadd, uint32_t, 2), ENTRY(dot4_plus_sadd, uint32_t, 2), ENTRY(dot4_plus_2salu, uint32_t, 3), ENTRY(fma64_plus_sadd, double, 2), ENTRY(dot4_plus_snop, uint32_t, 2),

    };
    printf("# %s, %d CUs, clock attribute %d kHz; %d loop iterations x 64 instructions per wave\n", prop.gcnArchName, ncu, clk, iters);
    printf("# per instruction and W (waves per SIMD; 'dep' = ONE dependent chain, one wave per SIMD):\n");
    printf("#   cyc  = s_memtime ticks of the median wave / (instructions per wave x W)   [ticks per wave-instruction per SIMD]\n");
    printf("#   ns   = hipEvent time of the launch / (instructions per wave x W)          [ns per wave-instruction per SIMD, whole chip]\n");
    printf("#   (ns is the figure to trust across W: the clock the chip holds differs between instruction mixes and loads)\n");
    printf("%-15s", "instruction");
    for (const char *w : { "W=1", "W=2", "W=4", "W=8", "dep" }) printf(" | %4s cyc    ns", w);
    printf("\n");
    bool hw_printed = false;
    for (const Entry &e : tab) {
        printf("%-15s", e.name);
        for (int c = 0; c < 5; c++) {
            const int wps = c < 4 ? (1 << c) : 1; const bool dep = c == 4;
            const int nw = ncu * 4 * wps;
            e.run(wps, dep, 50, out, hwid);                                     // warm (code object, clocks)
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            e.run(wps, dep, iters, out, hwid);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), out, nw * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hid.data(), hwid, nw * 4, hipMemcpyDeviceToHost));
            if (!hw_printed && c == 2) {
                // HW_ID: [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se
                int simd[4] = { 0, 0, 0, 0 }; const int wgw = 16; uint32_t cu0 = (hid[0] >> 8) & 0xff;
                bool same = true;
                for (int w = 0; w < wgw; w++) { simd[(hid[w] >> 4) & 3]++; same &= ((hid[w] >> 8) & 0xff) == cu0; }
                fprintf(stderr, "# placement of workgroup 0 at W=4 (16 waves): SIMD0..3 hold %d %d %d %d waves, one CU: %s\n", simd[0], simd[1], simd[2], simd[3], same ? "yes" : "NO");
                hw_printed = true;
            }
            std::sort(h.begin(), h.begin() + nw);
            const double ninstr = (double)iters * 64 * e.per_copy;
            printf(" | %8.2f %5.2f", h[nw / 2] / (ninstr * wps), ms * 1e6 / (ninstr * wps));
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

/* div_const_check: hvo_div_const (csrc/hvo_internal.hpp) against the IEEE division it replaces, on the CPU.
 * gcc -O2 -mfma -ffp-contract=off div_const_check.c -lm && ./a.out [samples per divisor, default 60000000]
 * Prints the number of operands whose quotient differs (two corrections: the shipped form; one correction: for the record). */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static inline uint64_t xs(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline double divc(double a, double b, double y)
{
    double q0 = a * y;
    double r0 = fma(-q0, b, a);
    double q1 = fma(r0, y, q0);
    double r1 = fma(-q1, b, a);
    return fma(r1, y, q1);
}
static inline double divc1(double a, double b, double y)
{
    double q0 = a * y;
    double r0 = fma(-q0, b, a);
    return fma(r0, y, q0);
}
int main(int argc, char **argv)
{
    const long per = argc > 1 ? atol(argv[1]) : 60000000;
    float bs[] = { 535.4f, 539.2f, 517.3f, 516.5f, 520.9f, 521.0f, 1070.8f, 1078.4f, 3.0f, 7.0f, 0.1f, 1e-3f, 123456.7f, 1.9999999f, 1.0000001f };
    long bad2 = 0, bad1 = 0, n = 0;
    for (unsigned k = 0; k < sizeof(bs) / sizeof(bs[0]); k++) {
        double b = (double)bs[k], y = 1.0 / b;
        for (long i = 0; i < per; i++) {
            uint64_t u = xs();
            double a;
            if (i & 1) { /* the path's own shape: (int - float) * (u16 * float) */
                int j = (int)(u % 1280); int d = (int)((u >> 16) & 0xFFFF); float c = 320.1f + (float)((u >> 40) & 255) * 0.01f;
                a = ((double)j - (double)c) * ((double)d * (double)(1.0f / 5000.0f));
            } else { uint64_t m = (u & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 20 + (u >> 58)) << 52); memcpy(&a, &m, 8); if (u & (1ull << 57)) a = -a; }
            double t = a / b;
            if (divc(a, b, y) != t) bad2++;
            if (divc1(a, b, y) != t) bad1++;
            n++;
        }
    }
    printf("n=%ld mismatches: two corrections %ld, one correction %ld\n", n, bad2, bad1);
    return bad2 != 0;
}

/* Exhaustive proof obligation of guardx_amd/csrc/gx_device.h:div_bin16 (test infrastructure; built and run by
 * tests/test_div_bin_size.py):   fma(fma(-(x*inv), bs, x), inv, x*inv) == x / bs   bit for bit, for every fp32 x with
 * 2^-100 <= |x| <= 2 pi (+ a margin), both signs, and for +0 -- with bs = fl(2 pi / 16), inv = fl(1 / bs).
 * Prints the number of inputs checked and of mismatches; also reports the first exponent at which the identity starts
 * to fail below the bound (so the bound in the kernel is known to be needed, not just sufficient). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void)
{
    const float bs = (float)((M_PI * 2) / 16); /* engine.py:880, gx_api.hip: p.bin_size */
    const float inv = 1.0f / bs;
    if (f2u(bs) != f2u(0x1.921fb6p-2f) || f2u(inv) != f2u(0x1.45f306p+1f)) { printf("constants differ\n"); return 2; }
    const uint32_t lo = 0x0D800000u;                          /* 2^-100: kDivFastMinBits */
    const uint32_t hi = f2u(6.2831854820251465f) + (1u << 20); /* 2 pi and a good margin above */
    long long bad = 0, n = 0, bad_below = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad, n, bad_below)
    for (int e = 0; e < 256; ++e) {
        for (uint32_t m = 0; m < (1u << 23); ++m) {
            const uint32_t b = ((uint32_t)e << 23) | m;
            if (b > hi) break;
            for (int sgn = 0; sgn < 2; ++sgn) {
                const float x = u2f(b | ((uint32_t)sgn << 31));
                const float q = x / bs, q0 = x * inv, r = fmaf(-q0, bs, x), q1 = fmaf(r, inv, q0);
                const int ne = f2u(q) != f2u(q1);
                if (b >= lo || (b == 0 && sgn == 0)) { n++; bad += ne; }
                else bad_below += ne;
            }
        }
    }
    printf("checked %lld mismatches %lld below_bound_mismatches %lld\n", n, bad, bad_below);
    return bad ? 1 : 0;
}

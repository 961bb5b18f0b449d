// rcp_exact_probe.hip -- for which x does a SHORT reciprocal sequence give the bits of the IEEE quotient 1.0f / x ?
//
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o rcp_exact_probe rcp_exact_probe.hip && ./rcp_exact_probe
//
// The kernels' solves are written in reciprocal form (gx_robot_ant.h:ldl_factor, arrow_solve: 1.0f / pivot, then
// multiplies), and the compiler turns every `1.0f / x` into the full IEEE division sequence: 2 v_div_scale, v_rcp,
// 5 v_fma / v_mul, v_div_fmas, v_div_fixup -- 11 dependent instructions, ~20 of them per Ant step on a one-wave-per-SIMD
// chain that pays 4+ cycles for every instruction it issues.  Candidates (all inputs, all 2^32 bit patterns, compared
// bit for bit with the compiler's IEEE sequence evaluated in the same kernel; NaN payloads compared as "both NaN"):
//   A: y0 = v_rcp_f32(x); e = fma(-x, y0, 1); y1 = fma(e, y0, y0); v_div_fixup(y1, x, 1)                  4 instructions
//   B: A's y1, then e1 = fma(-x, y1, 1); y2 = fma(e1, y1, y1); fixup                                      6 instructions
//   C: Markstein's final step on the numerator: q = y1; r = fma(-x, q, 1); q' = fma(r, y1, q); fixup       6 instructions
//   D: A without the fixup (what tanh_f uses, gx_device.h: its divisor is in [2, 6.6e7])                                3 instructions
//   E: 2 / x as 2 * D(x), compared with the IEEE quotient 2.0f / x (the numerator of tanh_f's division)
//   F: D, ACCEPTED only when its result is a normal number (one v_cmp_class): "F accepted" counts accepted results that
//      differ from the IEEE quotient, "F rejected" the inputs a caller would have to send through the IEEE sequence.
//      Result: 0 accepted results differ; rejected are exactly the inputs outside 2^-126 <= |x| <= 2^126 -- a bit-safe
//      short reciprocal exists.  The two forms of the Ant's / Walker's solves built on it (round 5) both lost to the
//      plain IEEE division in situ: profiles/r05_ab_short_rcp.log, DESIGN.md section 10.
// Reported per candidate: mismatches, and the binary exponents of x where they occur.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float rcp_A(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    return __builtin_amdgcn_div_fixupf(y1, x, 1.0f);
}
__device__ __forceinline__ float rcp_B(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float e1 = __builtin_fmaf(-x, y1, 1.0f);
    const float y2 = __builtin_fmaf(e1, y1, y1);
    return __builtin_amdgcn_div_fixupf(y2, x, 1.0f);
}
__device__ __forceinline__ float rcp_C(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float r = __builtin_fmaf(-x, y1, 1.0f);
    const float q = __builtin_fmaf(r, y0, y1);
    return __builtin_amdgcn_div_fixupf(q, x, 1.0f);
}

__device__ __forceinline__ float rcp_D(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, y0, 1.0f), y0, y0);
}

__device__ __forceinline__ bool same(float a, float b)
{
    if (a != a && b != b) return true;
    return __float_as_uint(a) == __float_as_uint(b);
}

// hist[c][e]: mismatches of candidate c among the inputs with biased exponent e (0 = zero / denormal, 255 = inf / NaN)
__global__ void probe(unsigned long long* hist, uint32_t* example)
{
    const uint64_t n = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += n) {
        const float x = __uint_as_float((uint32_t)b);
        const float ref = 1.0f / x;
        const int ex = (int)((b >> 23) & 255u);
        const float c[5] = {rcp_A(x), rcp_B(x), rcp_C(x), rcp_D(x), 2.0f * rcp_D(x)};
        const float refs[5] = {ref, ref, ref, ref, 2.0f / x};
        if (__builtin_isnormal(c[3])) { if (!same(c[3], ref)) atomicAdd(&hist[5 * 256 + ex], 1ull); }
        else atomicAdd(&hist[6 * 256 + ex], 1ull);
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (!same(c[k], refs[k])) {
                const unsigned long long was = atomicAdd(&hist[k * 256 + ex], 1ull);
                if (was == 0 && ex > 0 && ex < 255) { example[(k * 256 + ex) * 3] = (uint32_t)b; example[(k * 256 + ex) * 3 + 1] = __float_as_uint(c[k]); example[(k * 256 + ex) * 3 + 2] = __float_as_uint(refs[k]); }
            }
    }
}

int main()
{
    unsigned long long* d_hist; uint32_t* d_ex;
    CK(hipMalloc(&d_hist, sizeof(unsigned long long) * 7 * 256));
    CK(hipMalloc(&d_ex, sizeof(uint32_t) * 5 * 256 * 3));
    CK(hipMemset(d_hist, 0, sizeof(unsigned long long) * 7 * 256));
    CK(hipMemset(d_ex, 0, sizeof(uint32_t) * 5 * 256 * 3));
    hipLaunchKernelGGL(probe, dim3(8192), dim3(256), 0, 0, d_hist, d_ex);
    CK(hipDeviceSynchronize());
    static unsigned long long h[7 * 256]; static uint32_t ex[5 * 256 * 3];
    CK(hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ex, d_ex, sizeof ex, hipMemcpyDeviceToHost));
    const char* names[5] = {"A rcp+1NR+fixup (4 instr)", "B rcp+2NR+fixup (6 instr)", "C rcp+NR+residual-correction+fixup (6 instr)",
                            "D rcp+1NR, no fixup (3 instr)", "E 2*D(x) vs 2.0f/x"};
    for (int k = 0; k < 5; ++k) {
        unsigned long long tot = 0, normal = 0;
        int lo = 999, hi = -1;
        for (int e = 0; e < 256; ++e) {
            tot += h[k * 256 + e];
            if (e > 0 && e < 255 && h[k * 256 + e]) { normal += h[k * 256 + e]; if (e < lo) lo = e; if (e > hi) hi = e; }
        }
        printf("%s: %llu mismatches of 2^32 (zero/denormal inputs: %llu, inf/nan inputs: %llu, normal inputs: %llu",
               names[k], tot, h[k * 256], h[k * 256 + 255], normal);
        if (normal) printf("; biased exponents %d..%d", lo, hi);
        printf(")\n");
        for (int e = 1; e < 255; ++e)
            if (h[k * 256 + e]) {
                float x, c, r; memcpy(&x, &ex[(k * 256 + e) * 3], 4); memcpy(&c, &ex[(k * 256 + e) * 3 + 1], 4); memcpy(&r, &ex[(k * 256 + e) * 3 + 2], 4);
                printf("   exp %3d (2^%d): %llu   e.g. x=%a got %a want %a\n", e, e - 127, h[k * 256 + e], x, c, r);
            }
    }
    for (int k = 5; k < 7; ++k) {
        unsigned long long tot = 0, normal = 0;
        int lo = 999, hi = -1;
        for (int e = 0; e < 256; ++e) {
            tot += h[k * 256 + e];
            if (e > 0 && e < 255 && h[k * 256 + e]) { normal += h[k * 256 + e]; if (e < lo) lo = e; if (e > hi) hi = e; }
        }
        printf("%s: %llu of 2^32 (zero/denormal inputs: %llu, inf/nan inputs: %llu, normal inputs: %llu",
               k == 5 ? "F accepted (result a normal number) but not the IEEE quotient" : "F rejected (result not a normal number: the caller falls back)",
               tot, h[k * 256], h[k * 256 + 255], normal);
        if (normal) printf("; biased exponents %d..%d", lo, hi);
        printf(")\n");
    }
    return 0;
}

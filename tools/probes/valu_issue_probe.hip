// Issue cost of the integer VALU instructions a Threefry-2x32 round is made of, on gfx950, against the number of
// waves that share a SIMD.  Every test runs ITER x 32 instructions of one kind per wave (8 independent registers, so
// no dependency stalls; the "dep" variants chain through one register) and reports cycles per wave-instruction per
// SIMD = elapsed shader clocks x (100 MHz -> core clock is not known: wall time is used) ...
// Output: ns per wave-instruction per SIMD for W = 1, 2, 4, 8 waves per SIMD (all 256 CUs busy).
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue_probe tools/probes/valu_issue_probe.hip && ./valu_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(64) void probe(unsigned* out, int iters, unsigned seed)
{
    unsigned r[8], k = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = seed * (i + 1) + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) { // v_add_u32, independent
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 1) { // v_xor_b32
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 2) { // v_alignbit_b32 (rotate)
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %0, 19" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if (KIND == 3) { // v_add3_u32
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 4) { // v_xad_u32
#define X(i) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 5) { // v_lshl_or_b32
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 13, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 6) { // v_lshrrev_b32
#define X(i) asm volatile("v_lshrrev_b32 %0, 19, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if (KIND == 7) { // v_perm_b32 (rotate by 16 / 24)
#define X(i) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 8) { // v_bitop3_b32
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 9) { // v_add_u32 dependent chain
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[0]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 10) { // v_alignbit dependent chain
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %0, 19" : "+v"(r[0]));
                REP8(X)
#undef X
            } else if (KIND == 11) { // threefry round: add, alignbit, xor (dependent), two independent blocks
#define X(i) asm volatile("v_add_u32 %0, %0, %1\n v_alignbit_b32 %1, %1, %1, 19\n v_xor_b32 %1, %1, %0\n" \
                          "v_add_u32 %2, %2, %3\n v_alignbit_b32 %3, %3, %3, 19\n v_xor_b32 %3, %3, %2" \
                          : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
                REP8(X)
#undef X
            } else if (KIND == 12) { // v_mov_b32
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 13) { // v_alignbit with SGPR/inline shift in VOP3 form, distinct sources
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 19" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 14) { // v_mul_lo_u32 (rotate by multiply? no -- reference for a quarter-rate op)
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 15) { // v_pk_add_u16 (two 16-bit adds)
#define X(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            } else if (KIND == 16) { // v_lshl_add_u64
                unsigned long long* q = reinterpret_cast<unsigned long long*>(r);
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i & 3]) : "v"(q[(i + 1) & 3]));
                REP8(X)
#undef X
            } else if (KIND == 17) { // v_and_or_b32
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(k));
                REP8(X)
#undef X
            }
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= r[i];
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

typedef void (*kern_t)(unsigned*, int, unsigned);

int main()
{
    const char* names[] = {"v_add_u32", "v_xor_b32", "v_alignbit_b32 x,x", "v_add3_u32", "v_xad_u32", "v_lshl_or_b32",
                           "v_lshrrev_b32", "v_perm_b32", "v_bitop3_b32", "v_add_u32 dep chain", "v_alignbit dep chain",
                           "threefry round x2 (6 instr)", "v_mov_b32", "v_alignbit_b32 x,k", "v_mul_lo_u32",
                           "v_pk_add_u16", "v_lshl_add_u64", "v_and_or_b32"};
    kern_t ks[] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>,
                   probe<10>, probe<11>, probe<12>, probe<13>, probe<14>, probe<15>, probe<16>, probe<17>};
    const int per_iter[] = {32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 192, 32, 32, 32, 32, 32, 32};
    unsigned* out;
    hipMalloc(&out, 4096);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    printf("%-30s %10s %10s %10s %10s   (ns per wave-instruction per SIMD; 2.4 GHz: 2 cycles = 0.83 ns)\n", "instruction",
           "W=1", "W=2", "W=4", "W=8");
    for (int k = 0; k < 18; ++k) {
        printf("%-30s", names[k]);
        for (int W = 1; W <= 8; W *= 2) {
            const int grid = 256 * 4 * W; // one-wave workgroups: W per SIMD when spread evenly
            hipLaunchKernelGGL(ks[k], dim3(grid), dim3(64), 0, 0, out, 100, 1u);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(ks[k], dim3(grid), dim3(64), 0, 0, out, iters, 1u);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double n = (double)iters * per_iter[k] * W; // wave-instructions per SIMD
            printf(" %10.3f", ms * 1e6 / n);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

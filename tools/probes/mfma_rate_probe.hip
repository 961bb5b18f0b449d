// mfma_rate_probe.hip -- how many shader cycles does v_mfma_f32_16x16x4_f32 occupy the matrix pipe of a SIMD?
//
//   hipcc -O3 --offload-arch=gfx950 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
//
// One wave per SIMD (4-wave workgroups, one workgroup per CU), NCH independent accumulator chains issued round-robin,
// 4096 MFMAs per chain, stamped with s_memtime (shader clock) and s_memrealtime (100 MHz).  The closed-loop policy
// kernels (gx_policy.h) cost ~29 ns per MFMA on top of the width-64 step whatever feeds the operands; this separates the
// instruction's own cost from the kernels' surroundings.  Also timed: v_mfma_f32_32x32x2_f32 (same flops per instruction x 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* st, int iters)
{
    f4 acc[NCH];
    const float a = 1.0f + threadIdx.x * 1e-7f;
    float b[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { acc[c] = f4{0.f, 0.f, 0.f, 0.f}; b[c] = 0.5f + c * 1e-3f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

template <int NCH>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* st, int iters)
{
    f16v acc[NCH];
    const float a = 1.0f + threadIdx.x * 1e-7f;
    float b[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { for (int j = 0; j < 16; ++j) acc[c][j] = 0.f; b[c] = 0.5f + c * 1e-3f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[c], acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; st[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

template <class K>
static int run(const char* name, K kern, int nch, int wgs, int flops_per_mfma)
{
    float* out; unsigned long long* st;
    const int iters = 512; // x 8 x nch MFMAs per wave
    CK(hipMalloc(&out, sizeof(float) * wgs * 256));
    CK(hipMalloc(&st, sizeof(unsigned long long) * wgs * 8));
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, out, st, iters); }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(wgs * 8);
    CK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * wgs * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ns;
    const double n = (double)iters * 8 * nch;
    for (int w = 0; w < wgs * 4; ++w) { cyc.push_back(h[2 * w] / n); ns.push_back(h[2 * w + 1] * 10.0 / n); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ns.begin(), ns.end());
    printf("%-34s %3d workgroups x 4 waves, %d chains: %6.1f s_memtime ticks, %6.2f ns per MFMA per wave (median) = %6.1f TFLOP/s on 256 CUs\n",
           name, wgs, nch, cyc[cyc.size() / 2], ns[ns.size() / 2], flops_per_mfma / ns[ns.size() / 2] * 4 * 256 / 1e3);
    (void)hipFree(out); (void)hipFree(st);
    return 0;
}

int main()
{
    if (run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 125, 2048)) return 1;
    if (run("v_mfma_f32_16x16x4_f32", k16<8>, 8, 125, 2048)) return 1;
    if (run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 256, 2048)) return 1;
    if (run("v_mfma_f32_16x16x4_f32 (2 waves/SIMD)", k16<4>, 4, 512, 2048)) return 1;
    if (run("v_mfma_f32_32x32x2_f32", k32<2>, 2, 125, 4096)) return 1;
    if (run("v_mfma_f32_32x32x2_f32", k32<4>, 4, 256, 4096)) return 1;
    return 0;
}

// chain_clock_probe.hip -- what slows a lone serial chain when the rest of the chip is busy?  
//
//   hipcc -O3 --offload-arch=gfx950 -o chain_clock_probe chain_clock_probe.hip && ./chain_clock_probe
//
// The chain: 32 one-wave workgroups, each a dependent v_fma_f32 chain of fixed length (the shape of the dynamics pass of
// the two-kernel rollout), stamped with s_memtime (shader clock) and s_memrealtime (constant 100 MHz) at both ends.
// The filler: a kernel that keeps every SIMD it may use busy with independent FMAs (or with streaming loads / stores)
// for longer than the chain runs.  Streams with CU masks keep the two on DISJOINT compute units (the chain on the first
// four CUs of every XCD-sized group of the mask, the filler on the rest), so whatever the filler does to the chain does
// not come from sharing a CU.  Reported per case: the chain's duration in real time and in shader-clock ticks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void chain_kernel(int len, float* out, unsigned long long* stamps)
{
    extern __shared__ float lds_reserve[]; // size chosen by the launch: 0, or nearly a whole CU's LDS (nothing that needs LDS fits beside it)
    if (len < 0) lds_reserve[threadIdx.x] = 1.f;
    float x = 1.0f + threadIdx.x * 1e-6f, y = 0.999999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int k = 0; k < len; k += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, y, 1e-7f); // dependent: one v_fma_f32 per 4-cycle issue slot at best
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t1 - t0;
        stamps[blockIdx.x * 4 + 1] = r1 - r0;
        stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

// mode 0: independent FMAs (8 accumulators per lane); mode 1: streaming copy
__global__ __launch_bounds__(256) void filler_kernel(int mode, int iters, float* buf, size_t n, unsigned* cu_bitmap)
{
    if (threadIdx.x == 0) { // which CUs does the filler really run on?
        const unsigned key = ((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) << 7) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0x7fu);
        atomicOr(cu_bitmap + (key >> 5), 1u << (key & 31));
    }
    if (mode == 0) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = threadIdx.x * 1e-3f + u;
#pragma unroll 1
        for (int k = 0; k < iters; ++k) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = __builtin_fmaf(a[u], 0.999f, 0.5f);
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
        if (s == 12345.678f) buf[0] = s;
    } else if (mode == 2) { // the same arithmetic from ~96 KB of straight-line code: instruction fetch pressure
        __shared__ float tag[64];
        tag[threadIdx.x & 63] = 0.f; // (uses LDS: cannot share a CU with a chain workgroup that reserved it)
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = threadIdx.x * 1e-3f + u;
#pragma unroll 1
        for (int k = 0; k < iters; k += 1536) {
#pragma unroll
            for (int j = 0; j < 1536; ++j) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = __builtin_fmaf(a[u], 0.999f + j * 1e-9f, 0.5f);
            }
        }
        float s = tag[0];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
        if (s == 12345.678f) buf[0] = s;
    } else if (mode == 3) { // small code, but uses LDS (placement as mode 2)
        __shared__ float tag[64];
        tag[threadIdx.x & 63] = 0.f;
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = threadIdx.x * 1e-3f + u;
#pragma unroll 1
        for (int k = 0; k < iters; ++k) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = __builtin_fmaf(a[u], 0.999f, 0.5f);
        }
        float s = tag[0];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
        if (s == 12345.678f) buf[0] = s;
    } else {
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        for (int k = 0; k < iters; ++k)
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + n / 2 < n; i += stride) buf[i + n / 2] = buf[i] + 1.0f;
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
    // CU masks: bit i = CU i (HIP's enumeration).  Chain: every 8th CU (32 of 256); filler: all the others.
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> m_chain(words, 0), m_fill(words, 0), m_all(words, 0);
    for (int c = 0; c < ncu; ++c) {
        m_all[c / 32] |= 1u << (c % 32);
        if (c % 8 == 0) m_chain[c / 32] |= 1u << (c % 32); else m_fill[c / 32] |= 1u << (c % 32);
    }
    hipStream_t s_chain, s_fill, s_chain_all, s_fill_all;
    CK(hipExtStreamCreateWithCUMask(&s_chain, words, m_chain.data()));
    CK(hipExtStreamCreateWithCUMask(&s_fill, words, m_fill.data()));
    CK(hipStreamCreate(&s_chain_all));
    CK(hipStreamCreate(&s_fill_all));
    float *out, *buf; unsigned long long* stamps; unsigned* bitmap;
    CK(hipMalloc(&bitmap, 32 * 4));
    const size_t n = 256u << 20; // 1 GB of floats for the streaming filler
    CK(hipMalloc(&out, 32 * 64 * 4)); CK(hipMalloc(&stamps, 32 * 4 * 8)); CK(hipMalloc(&buf, n * 4));
    CK(hipMemset(buf, 0, n * 4));
    const int len = 200 * 1024; // ~100 us at one dependent FMA per ~5 cycles
    unsigned long long h[32 * 4];
    auto report = [&](const char* what) {
        (void)hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
        unsigned bm[32]; (void)hipMemcpy(bm, bitmap, sizeof bm, hipMemcpyDeviceToHost); (void)hipMemset(bitmap, 0, sizeof bm);
        int fill_cus = 0; for (int k = 0; k < 32; ++k) fill_cus += __builtin_popcount(bm[k]);
        int shared = 0;
        for (int w = 0; w < 32; ++w) { const unsigned key = (unsigned)(((h[w * 4 + 3] & 7) << 7) | ((h[w * 4 + 2] >> 8) & 0x7f)); shared += (bm[key >> 5] >> (key & 31)) & 1; }
        printf("[filler on %3d CUs, %2d chain waves on a filler CU] ", fill_cus, shared);
        std::vector<double> tk, rt;
        int cus = 0; std::vector<unsigned> seen;
        for (int w = 0; w < 32; ++w) {
            tk.push_back((double)h[w * 4]); rt.push_back((double)h[w * 4 + 1] * 0.01);
            const unsigned key = (unsigned)((h[w * 4 + 3] & 0xf) << 16 | (h[w * 4 + 2] & 0x7f00));
            if (std::find(seen.begin(), seen.end(), key) == seen.end()) { seen.push_back(key); ++cus; }
        }
        std::sort(tk.begin(), tk.end()); std::sort(rt.begin(), rt.end());
        printf("%-58s chain: %7.1f us real (max %7.1f), %8.0f shader ticks median -> %5.2f ticks/ns, %4.2f ticks per FMA; on %d CUs\n",
               what, rt[16], rt[31], tk[16], tk[16] / (rt[16] * 1e3), tk[16] / len, cus);
    };
    CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    for (int rep = 0; rep < 2; ++rep) {
        const size_t reserve = 160 * 1024 - 256;
        for (int mode = 3; mode >= 2; --mode) {
            hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(64), reserve, s_chain_all, len, out, stamps);   // the chain first: it owns its CUs' LDS
            hipLaunchKernelGGL(filler_kernel, dim3(ncu * 8), dim3(256), 0, s_fill_all, mode, 400000, buf, n, bitmap);
            CK(hipDeviceSynchronize());
            report(mode == 3 ? "chain reserves its CUs' LDS; VALU filler, small code" : "chain reserves its CUs' LDS; VALU filler, 96 KB of code");
        }
        // warm the clocks
        hipLaunchKernelGGL(filler_kernel, dim3(ncu * 8), dim3(256), 0, s_fill_all, 0, 200000, buf, n, bitmap);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(64), 0, s_chain_all, len, out, stamps);
        CK(hipDeviceSynchronize()); report("alone (no CU mask)");
        hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(64), 0, s_chain, len, out, stamps);
        CK(hipDeviceSynchronize()); report("alone (masked to 32 CUs)");
        for (int mode = 0; mode < 2; ++mode) {
            const int iters = mode == 0 ? 400000 : 2;
            // same CUs allowed for both (what the epoch does): the filler first, so that it is resident when the chain starts
            hipLaunchKernelGGL(filler_kernel, dim3(ncu * 8), dim3(256), 0, s_fill_all, mode, iters, buf, n, bitmap);
            hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(64), 0, s_chain_all, len, out, stamps);
            CK(hipDeviceSynchronize()); report(mode == 0 ? "beside a VALU filler, CUs shared" : "beside a streaming filler, CUs shared");
            // disjoint CUs
            hipLaunchKernelGGL(filler_kernel, dim3(ncu * 8), dim3(256), 0, s_fill, mode, iters, buf, n, bitmap);
            hipLaunchKernelGGL(chain_kernel, dim3(32), dim3(64), 0, s_chain, len, out, stamps);
            CK(hipDeviceSynchronize()); report(mode == 0 ? "beside a VALU filler on the OTHER 224 CUs" : "beside a streaming filler on the OTHER 224 CUs");
        }
    }
    return 0;
}

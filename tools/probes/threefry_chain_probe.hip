// ns per Threefry-2x32 block per SIMD for the sampler's `rng, rng1 = split(rng)` chain (the real device code of
// gx_device.h), against the waves per SIMD and the workgroup size.  The cost model from valu_issue_probe (VOP2 2
// cycles, VOP3 4 cycles) gives ~181 cycles per block.
//   hipcc --offload-arch=gfx950 -O3 -I guardx_amd/csrc -I include -o tools/probes/threefry_chain_probe tools/probes/threefry_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "gx_device.h"

template <int BLOCK, int ILP>
__global__ __launch_bounds__(BLOCK) void chain(unsigned* out, int links)
{
    uint32_t r0[ILP], r1[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) { r0[k] = blockIdx.x * BLOCK + threadIdx.x + k * 77777u; r1[k] = 12345u + k; }
    uint32_t acc = 0;
    for (int t = 0; t < links; ++t) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            uint32_t n0, n1, g0, g1;
            gx::split2(r0[k], r1[k], n0, n1, g0, g1);
            r0[k] = n0; r1[k] = n1; acc ^= g0 ^ g1;
        }
    }
    if (acc == 0x1234567u) out[threadIdx.x] = acc;
}

template <int BLOCK, int ILP>
static void run(unsigned* out, const char* name)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    printf("%-34s", name);
    const int links = 4000 / ILP;
    for (int W = 1; W <= 8; W *= 2) {
        const int grid = 256 * 4 * W * 64 / BLOCK;
        hipLaunchKernelGGL((chain<BLOCK, ILP>), dim3(grid), dim3(BLOCK), 0, 0, out, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((chain<BLOCK, ILP>), dim3(grid), dim3(BLOCK), 0, 0, out, links);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        printf(" %9.1f", ms * 1e6 / ((double)links * ILP * 2 * W));
    }
    printf("\n");
}

int main()
{
    unsigned* out;
    (void)hipMalloc(&out, 4096);
    printf("%-34s %9s %9s %9s %9s   (ns per block per SIMD; waves per SIMD W)\n", "kernel", "W=1", "W=2", "W=4", "W=8");
    run<64, 1>(out, "64-thread WG, 1 chain per lane");
    run<256, 1>(out, "256-thread WG, 1 chain per lane");
    run<64, 2>(out, "64-thread WG, 2 chains per lane");
    run<256, 2>(out, "256-thread WG, 2 chains per lane");
    run<64, 4>(out, "64-thread WG, 4 chains per lane");
    return 0;
}

// How many one-wave workgroups with L bytes of dynamic LDS are resident per CU on gfx950?  Every workgroup spins
// for a fixed number of shader-clock ticks; G = 256 * K workgroups finish in one spin time when K fit per CU and in
// two when they do not.
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/lds_residency_probe tools/probes/lds_residency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void spin(unsigned* out, long long ticks)
{
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 0xffffffffu) out[0] = 1;
}

int main()
{
    unsigned* out;
    (void)hipMalloc(&out, 64);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const long long ticks = 20000; // 100 MHz counter: 200 us
    printf("spin time per workgroup ~200 us; entries: elapsed us for K workgroups per CU\n%8s", "LDS B");
    const int Ks[] = {8, 12, 14, 15, 16, 18, 20, 24, 28, 32};
    for (int K : Ks) printf(" %6d", K);
    printf("\n");
    for (int L = 4096; L <= 12288; L += 512) {
        printf("%8d", L);
        for (int K : Ks) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), L, 0, out, 100);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a, 0);
            hipLaunchKernelGGL(spin, dim3(256 * K), dim3(64), L, 0, out, ticks);
            (void)hipEventRecord(b, 0);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            printf(" %6.0f", ms * 1000);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

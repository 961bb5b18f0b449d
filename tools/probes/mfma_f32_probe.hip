// Is v_mfma_f32_16x16x4_f32 (D = A*B + C, K = 4) bit-identical to a sequential fmaf chain over k,
// also when chained over several K-blocks?  Prints the number of mismatching outputs.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o mfma_probe tools/probes/mfma_f32_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void mfma_kernel(const float* A, const float* B, const float* C, float* D, int K)
{
    // A: [16][K] row-major, B: [K][16], C/D: [16][16]
    const int lane = threadIdx.x;
    const int i = lane & 15, kq = lane >> 4;
    f4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * kq + r) * 16 + i]; // lane holds rows 4*kq..+3 of column i
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float a = A[i * K + k0 + kq];       // A operand: lane = k*16 + row
        const float b = B[(k0 + kq) * 16 + i];    // B operand: lane = k*16 + col
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + i] = acc[r];
}

int main()
{
    const int K = 44;
    std::vector<float> A(16 * K), B(K * 16), C(256), D(256), R(256);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : A) v = rnd() * 3.f;
    for (auto& v : B) v = rnd();
    for (auto& v : C) v = rnd();
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, K);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad_seq = 0, bad_any = 0;
    double maxrel = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float acc = C[i * 16 + j];
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
            R[i * 16 + j] = acc;
            if (acc != D[i * 16 + j]) bad_seq++;
            double rel = fabs((double)acc - D[i * 16 + j]) / (fabs((double)acc) + 1e-30);
            if (rel > maxrel) maxrel = rel;
            if (rel > 1e-5) bad_any++;
        }
    printf("K=%d  mismatches vs sequential fmaf chain: %d / 256   (gross errors: %d, max rel diff %.3g)\n", K,
           bad_seq, bad_any, maxrel);
    return 0;
}

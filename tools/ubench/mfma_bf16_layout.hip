// Operand / result layout of v_mfma_f32_32x32x16_bf16 on gfx950, checked against a host product (gemm_mfma.h relies on it):
//   A: lane l holds A[row = l & 31][k = 8 * (l >> 5) .. + 7];  B: lane l holds B[k = 8 * (l >> 5) .. + 7][col = l & 31];
//   C: acc[reg] = C[row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)][col = l & 31].
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_bf16_layout.hip -o tools/ubench/mfma_bf16_layout && tools/ubench/mfma_bf16_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *C) {  // A [32][16], B [16][32] (values exactly representable in bf16)
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) C[((reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[reg];
}
int main() {
    float hA[512], hB[512], hC[1024], *dA, *dB, *dC;
    srand(1);
    for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 17 - 8); hB[i] = (float)(rand() % 13 - 6); }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float s = 0.f;
            for (int kk = 0; kk < 16; ++kk) s += hA[m * 16 + kk] * hB[kk * 32 + n];
            bad += s != hC[m * 32 + n];
        }
    printf("mfma_f32_32x32x16_bf16 layout: %s (%d of 1024 elements differ)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}

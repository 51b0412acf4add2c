// Micro-benchmark: issue rate of v_sqrt_f32 / v_rsq_f32 / v_pk_fma_f32 / v_fma_f32 on gfx950 (cycles per wave64 instruction
// and SIMD with the SIMDs saturated: 16 independent chains per wave, 8 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o trans_rate trans_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(float *out, const float *in, int iters) {
    float a[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + i] + 1.5f;
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    const float c = in[0] + 0.999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (OP == 0) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if constexpr (OP == 1) a[i] = __builtin_amdgcn_rsqf(a[i]);
            if constexpr (OP == 2) a[i] = __builtin_fmaf(a[i], c, c);
            if constexpr (OP == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
        }
        if constexpr (OP == 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], f2{c, c}, f2{c, c});
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, float *out, float *in) {
    const int iters = 4000, blocks = 256 * 4;  // 4 workgroups of 8 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, in, 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, in, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 16 * 8;  // 16 instructions per iteration and wave, 8 waves per SIMD
    printf("%-14s %8.3f ms  %.2f ns per wave instruction and SIMD  (= %.2f cycles @2.4 GHz)\n", name, ms, ms * 1e6 / insts_per_simd,
           ms * 1e6 / insts_per_simd * 2.4);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMalloc(&in, 4096 * 4);
    hipMemset(in, 0, 4096 * 4);
    run<2>("v_fma_f32", out, in);
    run<3>("v_pk_fma_f32", out, in);
    run<0>("v_sqrt_f32", out, in);
    run<1>("v_rsq_f32", out, in);
    run<4>("v_rcp_f32", out, in);
    return 0;
}

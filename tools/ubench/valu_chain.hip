// Micro-benchmark: cost of the RotatE pair term (backward and forward bodies) per wave on gfx950, in the
// shapes the pooled kernels use.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_chain valu_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct C { float re, im; };
typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(1024) void k(float *out, const float *in, int iters, unsigned mask) {
    float q0[8], q1[8], a0[8], a1[8], p0[8], p1[8];
    for (int r = 0; r < 8; ++r) { q0[r] = in[threadIdx.x + r]; q1[r] = in[threadIdx.x + 8 + r]; a0[r] = 0; a1[r] = 0; p0[r] = 0; p1[r] = 0; }
    float x0 = in[threadIdx.x + 100], x1 = in[threadIdx.x + 101];
    float g[8];
    for (int r = 0; r < 8; ++r) g[r] = in[r + 200];
    float s = 0.f;
    const unsigned m = __builtin_amdgcn_readfirstlane(mask);
    for (int it = 0; it < iters; ++it) {
        x0 += 1e-3f; x1 -= 1e-3f;
        if constexpr (V == 0) {  // backward body, uniform branches (as shipped)
#pragma unroll
            for (int r = 0; r < 8; ++r) if (m & (1u << r)) {
                float a = q0[r] - x0, b = q1[r] - x1; float n2 = a * a + b * b;
                float w = g[r] * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-30f));
                a0[r] += w * a; a1[r] += w * b;
            }
        } else if constexpr (V == 1) {  // backward body, no branches
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = q0[r] - x0, b = q1[r] - x1; float n2 = a * a + b * b;
                float w = g[r] * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-30f));
                a0[r] += w * a; a1[r] += w * b;
            }
        } else if constexpr (V == 2) {  // forward body, uniform branches
#pragma unroll
            for (int r = 0; r < 8; ++r) if (m & (1u << r)) {
                float a = q0[r] - x0, b = q1[r] - x1;
                a0[r] += __builtin_amdgcn_sqrtf(a * a + b * b);
            }
        } else if constexpr (V == 3) {  // forward body, no branches
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = q0[r] - x0, b = q1[r] - x1;
                a0[r] += __builtin_amdgcn_sqrtf(a * a + b * b);
            }
        } else if constexpr (V == 4) {  // 8 independent fma chains (VALU peak reference): 4 fma per r
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                a0[r] = fmaf(a0[r], x0, q0[r]); a1[r] = fmaf(a1[r], x1, q1[r]);
                a0[r] = fmaf(a0[r], x1, q1[r]); a1[r] = fmaf(a1[r], x0, q0[r]);
            }
        } else if constexpr (V == 5) {  // rsq only: 8 per iteration
#pragma unroll
            for (int r = 0; r < 8; ++r) a0[r] = __builtin_amdgcn_rsqf(a0[r] + x0);
        } else if constexpr (V == 7) {  // backward body as shipped: two units per lane, packed math (7 pk + 2 rsq per row)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f2 a = f2{q0[r], q1[r]} - f2{x0, x1}, b = f2{q1[r], q0[r]} - f2{x1, x0};
                f2 n2 = __builtin_elementwise_fma(b, b, __builtin_elementwise_fma(a, a, f2{1e-30f, 1e-30f}));
                f2 w = f2{__builtin_amdgcn_rsqf(n2.x), __builtin_amdgcn_rsqf(n2.y)} * g[r];
                f2 c0 = f2{a0[r], a1[r]} - w * a, c1 = f2{p0[r], p1[r]} - w * b;
                a0[r] = c0.x; a1[r] = c0.y; p0[r] = c1.x; p1[r] = c1.y;
            }
        } else if constexpr (V == 8) {  // forward body as shipped: two units per lane (5 pk + 2 sqrt per row)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f2 a = f2{q0[r], q1[r]} - f2{x0, x1}, b = f2{q1[r], q0[r]} - f2{x1, x0};
                f2 n2 = a * a + b * b;
                f2 c0 = f2{a0[r], a1[r]} + f2{__builtin_amdgcn_sqrtf(n2.x), __builtin_amdgcn_sqrtf(n2.y)};
                a0[r] = c0.x; a1[r] = c0.y;
            }
        } else if constexpr (V == 9) {  // packed fma peak reference: 4 pk_fma per row
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f2 c = f2{a0[r], a1[r]}, d = f2{p0[r], p1[r]};
                c = c * f2{x0, x1} + f2{q0[r], q1[r]}; d = d * f2{x1, x0} + f2{q1[r], q0[r]};
                c = c * f2{x1, x0} + f2{q1[r], q0[r]}; d = d * f2{x0, x1} + f2{q0[r], q1[r]};
                a0[r] = c.x; a1[r] = c.y; p0[r] = d.x; p1[r] = d.y;
            }
        } else if constexpr (V == 6) {  // backward body without rsq (mul instead) - isolates the transcendental
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = q0[r] - x0, b = q1[r] - x1; float n2 = a * a + b * b;
                float w = g[r] * fmaxf(n2, 1e-30f);
                a0[r] += w * a; a1[r] += w * b;
            }
        }
    }
    for (int r = 0; r < 8; ++r) s += a0[r] + a1[r] + p0[r] + p1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(const char *name, int blocks, int threads, int iters, float ops_per_iter) {
    float *out, *in;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&in, sizeof(float) * 4096);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 97) + 0.01f;
    hipMemcpy(in, h.data(), sizeof(float) * 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, 0xffu);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, in, iters, 0xffu);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks*threads/64 / 1024 SIMDs
    double waves_per_simd = (double)blocks * threads / 64.0 / 1024.0;
    double ns_per_pair = ms * 1e6 / (iters * 8.0 * waves_per_simd);  // SIMD-time per (wave, row) body
    printf("%-44s blocks=%4d thr=%4d  %8.3f ms  %7.2f ns per wave-pair  (~%5.1f cycles @2.1GHz)\n", name, blocks, threads, ms,
           ns_per_pair, ns_per_pair * 2.1);
    hipFree(out); hipFree(in);
}

int main() {
    const int it = 20000;
    for (int thr : {1024, 256}) {
        int blocks = thr == 1024 ? 256 : 1024;  // 4 waves / SIMD in both
        run<0>("bwd body, uniform branches", blocks, thr, it, 0);
        run<1>("bwd body, straight line", blocks, thr, it, 0);
        run<2>("fwd body, uniform branches", blocks, thr, it, 0);
        run<3>("fwd body, straight line", blocks, thr, it, 0);
        run<4>("4 fma per row (peak reference)", blocks, thr, it, 0);
        run<5>("1 rsq per row", blocks, thr, it, 0);
        run<6>("bwd body without rsq", blocks, thr, it, 0);
    }
    for (int wps : {1, 2, 4}) {
        printf("-- packed bodies (two units per lane), %d waves / SIMD\n", wps);
        run<7>("bwd body packed (7 pk + 2 rsq)", 128 * wps, 512, it, 0);
        run<8>("fwd body packed (5 pk + 2 sqrt)", 128 * wps, 512, it, 0);
        run<9>("4 pk_fma per row (packed peak reference)", 128 * wps, 512, it, 0);
        run<5>("1 rsq per row", 128 * wps, 512, it, 0);
    }
    run<1>("bwd straight, 8 waves/SIMD", 512, 1024, it, 0);
    run<1>("bwd straight, 2 waves/SIMD", 128, 1024, it, 0);
    run<1>("bwd straight, 1 wave/SIMD", 64, 1024, it, 0);
    return 0;
}

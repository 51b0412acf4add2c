// Microbenchmark (not part of the library): what a launch of 256 fat workgroups (1024 lanes, ~128 VGPRs, 130 KB of LDS: the shape
// of pool_bwd1) costs before its first useful instruction -- every wave stamps s_memtime at entry and after the first barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(1024) void fat_kernel(unsigned long long *out, int touch_lds) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    float keep[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) keep[i] = (float)(threadIdx.x + i);
    if (touch_lds)
        for (int e = threadIdx.x * 4; e < 32768; e += 4096) *reinterpret_cast<float4 *>(lds + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += keep[i] * (float)t1;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
        out[3 * w] = t0; out[3 * w + 1] = t1; out[3 * w + 2] = (unsigned long long)s;
    }
}
extern "C" int fat_launch(void *out, int touch, int grid, int lds_bytes, void *stream) {
    static bool once = false;
    if (!once) { hipFuncSetAttribute((const void *)fat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); once = true; }
    hipLaunchKernelGGL(fat_kernel, dim3(grid), dim3(1024), lds_bytes, (hipStream_t)stream, (unsigned long long *)out, touch);
    return (int)hipGetLastError();
}

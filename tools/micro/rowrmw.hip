// Microbenchmark (not part of the library): read-modify-write of RANDOM rows of a [N, D] fp32 table kept as 4 / 2 / 1 arrays
// (rows of 8 / 16 / 32 KB contiguous), one 512-lane workgroup per row -- what the row layout of the Adam state is worth.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(512) void rmw_kernel(float4 *a0, float4 *a1, float4 *a2, float4 *a3, int n_arrays,
                                                             int64_t row_vec4, const int64_t *ids) {
    const int64_t row = ids[blockIdx.x];
    float4 *arr[4] = {a0, a1, a2, a3};
    for (int64_t k = threadIdx.x; k < row_vec4; k += 512) {
        float4 v[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) if (a < n_arrays) v[a] = arr[a][row * row_vec4 + k];
#pragma unroll
        for (int a = 0; a < 4; ++a) if (a < n_arrays) { v[a].x += 1.f; v[a].y *= 0.5f; arr[a][row * row_vec4 + k] = v[a]; }
    }
}
extern "C" int rmw_launch(void *a0, void *a1, void *a2, void *a3, int n_arrays, int64_t row_vec4, const int64_t *ids, int n,
                          void *stream) {
    hipLaunchKernelGGL(rmw_kernel, dim3(n), dim3(512), 0, (hipStream_t)stream, (float4 *)a0, (float4 *)a1, (float4 *)a2,
                       (float4 *)a3, n_arrays, row_vec4, ids);
    return (int)hipGetLastError();
}

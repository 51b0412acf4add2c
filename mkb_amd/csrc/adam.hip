// mkb_adam_step: dense Adam with the exact element-wise semantics of torch.optim.Adam (single tensor,
// no amsgrad, no weight decay, maximize=False) as the reference's training loops use it
// (README.md:123-126; compose/pipeline.py:238-240), optionally fused with optimizer.zero_grad().
//
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//
// Pure HBM streaming: 4 reads + 3 writes (+1 write when zeroing g) of n floats; float4 per lane,
// grid-stride over <= 2048 workgroups.
#include "common.h"

#include <math.h>

namespace mkb {

// Same operation order as torch 2.x _single_tensor_adam on CPU so results track the reference to ~1 ulp:
//   exp_avg.lerp_(grad, 1-b1)  -> fma(1-b1, g-m, m)        (ATen lerp, weight < 0.5)
//   exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2) -> v*b2 + ((1-b2)*g)*g
//   denom = sqrt(v) / sqrt(bc2) + eps ;  p += (-step_size * m) / denom
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float w1, float b2, float w2,
                                         float neg_step, float sqrt_bc2, float eps) {
#pragma clang fp contract(off)
    m = fmaf(w1, g - m, m);
    v = v * b2;
    v = v + (w2 * g) * g;
    const float denom = sqrtf(v) / sqrt_bc2 + eps;
    p = p + (neg_step * m) / denom;
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, float w1, float b2, float w2, float neg_step,
                                                   float sqrt_bc2, float eps, int zero_grad) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g);
    float4 *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, w1, b2, w2, neg_step, sqrt_bc2, eps);
        adam_one(pp.y, gg.y, mm.y, vv.y, w1, b2, w2, neg_step, sqrt_bc2, eps);
        adam_one(pp.z, gg.z, mm.z, vv.z, w1, b2, w2, neg_step, sqrt_bc2, eps);
        adam_one(pp.w, gg.w, mm.w, vv.w, w1, b2, w2, neg_step, sqrt_bc2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, w1, b2, w2, neg_step, sqrt_bc2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad) g[i] = 0.f;
    }
}

}  // namespace mkb

extern "C" int mkb_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step,
                             float lr, float beta1, float beta2, float eps, int zero_grad, void *stream) {
    MKB_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
    MKB_REQUIRE(n >= 0 && step >= 1, "bad n / step");
    MKB_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "buffers must be 16-byte aligned");
    if (n == 0) return MKB_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float neg_step = (float)(-((double)lr / bc1));
    const float sqrt_bc2 = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    mkb::ProfScope ps(MKB_PROF_ADAM, (hipStream_t)stream);
    hipLaunchKernelGGL(mkb::adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, w1, beta2, w2, neg_step, sqrt_bc2, eps, zero_grad);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

// mkb_kl_divergence: distillation loss, forward AND both gradient seeds in one pass.
//
// Replaces losses.KlDivergence.__call__ (losses/kl_divergence.py:22-29):
//   loss = mean over ALL n*m entries of  F.kl_div(log_softmax(student / T, dim=1), softmax(teacher / T, dim=1), 'none')
//        = 1/(n m) * sum_i sum_j t_ij (log t_ij - log p_ij),     t = softmax(teacher / T), p = softmax(student / T)
//   (entries with t_ij == 0 contribute 0, as F.kl_div's xlogy does).  Closed-form gradients:
//   d loss / d student_ik = (p_ik - t_ik) / (n m T)
//   d loss / d teacher_ik = t_ik ((log t_ik - log p_ik) - KL_i) / (n m T),      KL_i = sum_j t_ij (log t_ij - log p_ij)
// One wave per row (the distributions of distillation/distillation.py:486-558 hold a few entities / relations, m is small);
// correctly rounded expf / logf: this loss is compared with torch's CPU result at 1e-6.  Row partials are summed in a fixed
// order by a second single-workgroup launch: bit-reproducible.
#include "common.h"
#include "model_math.h"

namespace mkb {

__global__ __launch_bounds__(256) void kl_rows_kernel(const float *__restrict__ student, const float *__restrict__ teacher, int n,
                                                      int m, float inv_T, float *__restrict__ dstudent,
                                                      float *__restrict__ dteacher, float *__restrict__ rowpart) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *s = student + (int64_t)i * m, *t = teacher + (int64_t)i * m;
    float ms = -INFINITY, mt = -INFINITY;
    for (int j = lane; j < m; j += 64) {
        ms = fmaxf(ms, s[j] * inv_T);
        mt = fmaxf(mt, t[j] * inv_T);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, off, 64));
        mt = fmaxf(mt, __shfl_xor(mt, off, 64));
    }
    float zs = 0.f, zt = 0.f;
    for (int j = lane; j < m; j += 64) {
        zs += expf(s[j] * inv_T - ms);
        zt += expf(t[j] * inv_T - mt);
    }
    zs = wave_sum(zs);
    zt = wave_sum(zt);
    const float ls = logf(zs), lt = logf(zt);
    float kl = 0.f;
    for (int j = lane; j < m; j += 64) {
        const float logp = s[j] * inv_T - ms - ls, logt = t[j] * inv_T - mt - lt;
        const float tt = expf(logt);
        kl += tt > 0.f ? tt * (logt - logp) : 0.f;
    }
    kl = wave_sum(kl);
    const float c = inv_T / ((float)n * (float)m);
    for (int j = lane; j < m; j += 64) {
        const float logp = s[j] * inv_T - ms - ls, logt = t[j] * inv_T - mt - lt;
        const float pp = expf(logp), tt = expf(logt);
        dstudent[(int64_t)i * m + j] = c * (pp - tt);
        if (dteacher) dteacher[(int64_t)i * m + j] = tt > 0.f ? c * tt * ((logt - logp) - kl) : 0.f;
    }
    if (lane == 0) rowpart[i] = kl;
}

__global__ __launch_bounds__(256) void kl_finish_kernel(const float *__restrict__ rowpart, int n, float scale, float *__restrict__ loss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += rowpart[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_kl_divergence(const float *student, const float *teacher, int64_t n, int64_t m, float T, float *loss,
                                 float *dstudent, float *dteacher, float *scratch, void *stream) {
    MKB_REQUIRE(student && teacher && loss && dstudent && scratch, "null pointer");
    MKB_REQUIRE(n > 0 && m > 0 && n <= INT32_MAX && m <= INT32_MAX, "bad n / m");
    MKB_REQUIRE(T > 0.f, "temperature must be positive");
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(MKB_PROF_LOSS, st);
    hipLaunchKernelGGL(kl_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, student, teacher, (int)n, (int)m, 1.f / T,
                       dstudent, dteacher, scratch);
    hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(256), 0, st, scratch, (int)n, 1.f / ((float)n * (float)m), loss);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

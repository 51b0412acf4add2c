// mkb_adam_step: dense Adam with the exact element-wise semantics of torch.optim.Adam (single tensor,
// no amsgrad, no weight decay, maximize=False) as the reference's training loops use it
// (README.md:123-126; compose/pipeline.py:238-240), optionally fused with optimizer.zero_grad().
//
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//
// Pure HBM streaming: 4 reads + 3 writes (+1 write when zeroing g) of n floats; float4 per lane,
// grid-stride over <= 2048 workgroups.
#include "common.h"
#include "sampler_draw.h"

#include <math.h>
#include <algorithm>

namespace mkb {

// Same operation order as torch 2.x _single_tensor_adam on CPU so results track the reference to ~1 ulp:
//   exp_avg.lerp_(grad, 1-b1)  -> fma(1-b1, g-m, m)        (ATen lerp, weight < 0.5)
//   exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2) -> v*b2 + ((1-b2)*g)*g
//   denom = sqrt(v) / sqrt(bc2) + eps ;  p += (-step_size * m) / denom     (sqrt and the two divisions at 1 ulp)
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float w1, float b2, float w2,
                                         float neg_step, float sqrt_bc2, float eps) {
#pragma clang fp contract(off)
    m = fmaf(w1, g - m, m);
    v = v * b2;
    v = v + (w2 * g) * g;
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sqrtf and '/' (~30 instructions per
    // element): the update is lr-sized, so the deviation from torch's result is ~1e-7 of 5e-5 per step.  The dense
    // and the row-lazy kernels share this code, which is what keeps them bit-identical to each other.
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(sqrt_bc2) + eps;
    p = p + (neg_step * m) * __builtin_amdgcn_rcpf(denom);
}

// dense Adam (+ zero_grad) over a grid-stride range; shared by adam_kernel and the extra workgroups of the row step
__device__ __forceinline__ void adam_dense_range(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                 float *__restrict__ v, int64_t n, int64_t first, int64_t stride, float w1,
                                                 float b2, float w2, float neg_step, float sqrt_bc2, float eps, int zero_grad) {
    const int64_t n4 = n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g);
    float4 *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    for (int64_t i = first; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, w1, b2, w2, neg_step, sqrt_bc2, eps);
        adam_one(pp.y, gg.y, mm.y, vv.y, w1, b2, w2, neg_step, sqrt_bc2, eps);
        adam_one(pp.z, gg.z, mm.z, vv.z, w1, b2, w2, neg_step, sqrt_bc2, eps);
        adam_one(pp.w, gg.w, mm.w, vv.w, w1, b2, w2, neg_step, sqrt_bc2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + first; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, w1, b2, w2, neg_step, sqrt_bc2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad) g[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, float w1, float b2, float w2, float neg_step,
                                                   float sqrt_bc2, float eps, int zero_grad) {
    adam_dense_range(p, g, m, v, n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, w1, b2, w2,
                     neg_step, sqrt_bc2, eps, zero_grad);
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

namespace mkb {
constexpr int kAdamMulti = 8;
struct AdamMultiArgs {
    float *p[kAdamMulti], *g[kAdamMulti], *m[kAdamMulti], *v[kAdamMulti];
    int64_t n[kAdamMulti];
    float neg_step[kAdamMulti], sqrt_bc2[kAdamMulti];
    int first_block[kAdamMulti + 1];  // tensor t owns blocks [first_block[t], first_block[t + 1])
    int n_tensors, zero_grad;
    float w1, b2, w2, eps;
    int has_draw;  // the LAST block draws the negative sampler's next pool (pool_draw_body), as in the row-lazy launches
    DrawArgs draw;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMultiArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_multi_draw[];  // only sized when a draw block rides
    if (A.has_draw && (int)blockIdx.x == A.first_block[A.n_tensors]) {
        pool_draw_body<256>(A.draw, lds_multi_draw);
        return;
    }
    int t = 0;
    while (t + 1 < A.n_tensors && (int)blockIdx.x >= A.first_block[t + 1]) ++t;  // (workgroup-uniform)
    const int64_t nb = A.first_block[t + 1] - A.first_block[t];
    adam_dense_range(A.p[t], A.g[t], A.m[t], A.v[t], A.n[t], ((int64_t)blockIdx.x - A.first_block[t]) * 256 + threadIdx.x, nb * 256, A.w1,
                     A.b2, A.w2, A.neg_step[t], A.sqrt_bc2[t], A.eps, A.zero_grad);
}
}  // namespace mkb

extern "C" int mkb_adam_step_multi(const mkb_adam_dense_t *tensors, int n_tensors, float lr, float beta1, float beta2, float eps,
                                   int zero_grad, mkb_sampler_t *draw_ahead, void *stream) {
    MKB_REQUIRE(tensors && n_tensors >= 1 && n_tensors <= mkb::kAdamMulti, "1..%d tensors", mkb::kAdamMulti);
    mkb::AdamMultiArgs A{};
    A.n_tensors = n_tensors; A.zero_grad = zero_grad; A.eps = eps; A.b2 = beta2;
    A.w1 = (float)(1.0 - (double)beta1); A.w2 = (float)(1.0 - (double)beta2);
    int blocks = 0;
    for (int t = 0; t < n_tensors; ++t) {
        const mkb_adam_dense_t &T = tensors[t];
        MKB_REQUIRE(T.param && T.grad && T.exp_avg && T.exp_avg_sq && T.n >= 0 && T.step >= 1, "bad tensor %d", t);
        MKB_REQUIRE((((uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq) & 15) == 0,
                    "buffers must be 16-byte aligned");
        A.p[t] = T.param; A.g[t] = T.grad; A.m[t] = T.exp_avg; A.v[t] = T.exp_avg_sq; A.n[t] = T.n;
        const double bc1 = 1.0 - pow((double)beta1, (double)T.step), bc2 = 1.0 - pow((double)beta2, (double)T.step);
        A.neg_step[t] = (float)(-((double)lr / bc1));  // (the same scalars, computed the same way, as mkb_adam_step: same bits)
        A.sqrt_bc2[t] = (float)sqrt(bc2);
        int64_t nb = ((T.n >> 2) + 255) / 256;
        if (nb < 1) nb = 1;
        if (nb > 2048) nb = 2048;
        A.first_block[t] = blocks;
        blocks += (int)nb;
    }
    A.first_block[n_tensors] = blocks;
    size_t lds = 0;
    if (draw_ahead && mkb::sampler_draw_ahead(draw_ahead, &A.draw, &lds)) { A.has_draw = 1; ++blocks; }
    mkb::ProfScope ps(MKB_PROF_ADAM, (hipStream_t)stream);
    hipLaunchKernelGGL(mkb::adam_multi_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

// =====================================================================================================
// Row-lazy dense Adam: identical arithmetic, HBM traffic proportional to the rows a step touches.
//
// Dense Adam moves EVERY row every step (rows with a zero gradient still decay their moments and drift along the
// stale momentum), which makes the optimizer the largest HBM consumer of a KGE step: 8 x 116 MB per step at the
// headline config for ~2,400 touched rows out of 14,541.  But a row's zero-gradient steps depend on nothing but the
// row itself and the per-step scalars (lr / bias corrections), so they can be DEFERRED: `last[row]` records the
// step a row is current through; before a step reads a row (it is in the batch's pool / heads / tails) the
// pending zero-gradient steps are replayed in registers (same operations, same order, same rounding as the dense
// kernel would have applied them), and after backward only the touched rows take the real step.  A flush replays
// everything that is pending (before evaluation, checkpointing, or any direct read of the table).
// Rows that were never touched have m = v = 0, for which a dense step is the identity: they are skipped.
// consts[s] = (-lr / (1 - beta1^s), sqrt(1 - beta2^s)) is recorded by the step kernel for later replays.
//
// "Advance" form (mkb_adam_rows_advance*): the REAL step is deferred as well.  A row the batch of step t touched was made
// current through t-1 before the forward pass, so after backward it is the one row state "current through t-1, gradient of
// step t in its gradient row".  Nothing forces that step to be applied before the row is next read: the replay of a row
// simply takes the row's gradient for its FIRST pending step (a zero row for rows that were not touched then: adam_one at
// g = 0 is the zero-gradient step, bit for bit) and clears it.  The separate step launch (a second pass over p, m, v of
// every touched row) disappears; the small dense tensor that rode it rides the next advance launch instead.

namespace mkb {

struct AdamRowArgs {
    float *p, *g, *m, *v;
    int32_t *last;
    float2 *consts;
    const int64_t *ids;  // rows to process (duplicates allowed), or null = all rows (flush)
    int64_t D;
    int32_t step;        // catch-up / flush: bring rows to `step`; step kernel: apply `step`
    float w1, b2, w2, eps, neg_step, sqrt_bc2;
    // step kernel only: a small DENSE tensor (the relation table) stepped by extra workgroups of the same launch
    int32_t n_ids;
    float *dp, *dg, *dm, *dv;
    int64_t dn;
    float d_neg_step, d_sqrt_bc2;
    // catch-up kernel only: block 0 draws the negative sampler's NEXT pool (pool_draw_body) when first_row_block is set,
    // the n_filter blocks behind it filter THIS batch's rows (filter_rows_body), and the row ids may be given as the
    // three segments pool | heads | tails instead of one array (the array is what the filter blocks are still writing)
    int32_t first_row_block;  // 1 with a draw block, else 0
    int32_t n_filter;
    DrawArgs draw;
    FilterArgs filt;
    const int64_t *seg_pool, *seg_sample;
    int32_t seg_P, seg_B;
    // advance form (catch-up kernel with g != null): a row's first replayed step takes its gradient row; step `step`
    // uses (neg_step, sqrt_bc2) from here (recorded into consts[step] by the launch), n_ids = number of row blocks,
    // the dense rider (dp ...) is stepped by the blocks behind them
    int32_t n_rows_listed;  // catch-up kernel: rows to visit (n_ids row blocks of rows_per_block rows each)
    int32_t rows_per_block; // power of two <= 16: kCatchThreads / rows_per_block lanes x (float4 | float2) cover one row
    int32_t vec4;           // rows are read as float4 per lane (D % 4 == 0, 16-byte aligned arrays), else float2 / scalar
    // row-sharded table (mkb_adam_rows_advance_sharded): the first own_n listed rows are GLOBAL entity ids of a table whose
    // row e lives on rank e % own_world at index e / own_world; entries another rank owns are skipped.  `ids` follows them.
    const int64_t *own_ids;
    int32_t own_n, own_world, own_rank;
    // sweep: the listed rows [n_batch_rows, n_rows_listed) are table rows sweep_start, sweep_start + 1, ... (mod n_table) --
    // a window that moves through the table once per kSweepPeriod steps, so that no row's pending list outgrows the period
    int32_t n_batch_rows, sweep_start;
    int64_t n_table;
};

__device__ __forceinline__ void adam_zero_grad_step(float &p, float &m, float &v, float w1, float b2, float neg_step,
                                                    float sqrt_bc2, float eps) {
#pragma clang fp contract(off)
    m = fmaf(w1, 0.f - m, m);  // the dense kernel's fmaf(w1, g - m, m) at g = 0
    v = v * b2;                // ... + (w2 * 0) * 0 adds +0
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sqrtf and '/' (~30 instructions per
    // element): the update is lr-sized, so the deviation from torch's result is ~1e-7 of 5e-5 per step.  The dense
    // and the row-lazy kernels share this code, which is what keeps them bit-identical to each other.
    const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(sqrt_bc2) + eps;
    p = p + (neg_step * m) * __builtin_amdgcn_rcpf(denom);
}

// workgroup size of the catch-up / advance kernel (the sampler's filter and draw blocks that ride it are sized by it too)
// 512 lanes: one 2000-float row per workgroup, two workgroups per CU (107 VGPRs): their phases -- ownership exchange, loads, replay,
// stores -- interleave.  Round 4, same box, 1024 -> 512 lanes: headline step 0.2273 -> 0.2242 ms, WN18RR 0.1302 -> 0.1262, YAGO3-10
// 0.1933 -> 0.1862 (256 lanes: better still for YAGO3-10's long replays, worse at the headline; tools/_kb_catch.sh)
#ifndef MKB_CATCH_THREADS
#define MKB_CATCH_THREADS 512
#endif
constexpr int kCatchThreads = MKB_CATCH_THREADS;
#ifndef MKB_REPLAY_UNROLL
#define MKB_REPLAY_UNROLL 4
#endif
constexpr int kReplayUnroll = MKB_REPLAY_UNROLL;  // pending zero-gradient steps replayed side by side
typedef float f2 __attribute__((ext_vector_type(2)));

// adam_one / adam_zero_grad_step on two elements at once (v_pk_* where the scalar code has v_*: the same IEEE operations,
// element-wise, so the results are the scalar ones bit for bit); inv_bc2 = v_rcp_f32(sqrt_bc2), taken once per step
__device__ __forceinline__ void adam_pair(f2 &p, f2 g, f2 &m, f2 &v, float w1, float b2, float w2, float neg_step, float inv_bc2,
                                          float eps) {
#pragma clang fp contract(off)
    m = __builtin_elementwise_fma(f2{w1, w1}, g - m, m);
    v = v * b2;
    v = v + (w2 * g) * g;
    const f2 denom = f2{__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)} * inv_bc2 + eps;
    p = p + (neg_step * m) * f2{__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
}

__device__ __forceinline__ void adam_pair_zero_grad(f2 &p, f2 &m, f2 &v, float w1, float b2, float neg_step, float inv_bc2,
                                                    float eps) {
#pragma clang fp contract(off)
    // (0 - m) as a source modifier: 0 - m and -m differ only for m = +-0, where w1 * (+-0) + m gives the same +0 / -0 sum
    m = __builtin_elementwise_fma(f2{w1, w1}, -m, m);
    v = v * b2;  // (+ (w2 * 0) * 0 adds +0 to a non-negative number)
    const f2 denom = f2{__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)} * inv_bc2 + eps;
    p = p + (neg_step * m) * f2{__builtin_amdgcn_rcpf(denom.x), __builtin_amdgcn_rcpf(denom.y)};
}

// EPL (2 or 4) consecutive elements of one row per lane; VEC: the row is aligned for one EPL-wide access per array
template <int EPL>
struct RowChunk {
    float p[EPL], m[EPL], v[EPL], g[EPL];
};

template <int EPL>
__device__ __forceinline__ void replay_load(const AdamRowArgs &A, int64_t row, int k, RowChunk<EPL> &c) {
    const float *p = A.p + row * A.D + k, *m = A.m + row * A.D + k, *v = A.v + row * A.D + k;
    const float *g = A.g ? A.g + row * A.D + k : nullptr;
#pragma unroll
    for (int e = 0; e < EPL; ++e) c.g[e] = 0.f;
    if (k + EPL <= A.D && (A.D % EPL) == 0) {
        if constexpr (EPL == 4) {
            const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(m),
                         d = *reinterpret_cast<const float4 *>(v);
            c.p[0] = a.x; c.p[1] = a.y; c.p[2] = a.z; c.p[3] = a.w;
            c.m[0] = b.x; c.m[1] = b.y; c.m[2] = b.z; c.m[3] = b.w;
            c.v[0] = d.x; c.v[1] = d.y; c.v[2] = d.z; c.v[3] = d.w;
            if (g) {
                const float4 h = *reinterpret_cast<const float4 *>(g);
                c.g[0] = h.x; c.g[1] = h.y; c.g[2] = h.z; c.g[3] = h.w;
            }
        } else {
            const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(m),
                         d = *reinterpret_cast<const float2 *>(v);
            c.p[0] = a.x; c.p[1] = a.y; c.m[0] = b.x; c.m[1] = b.y; c.v[0] = d.x; c.v[1] = d.y;
            if (g) {
                const float2 h = *reinterpret_cast<const float2 *>(g);
                c.g[0] = h.x; c.g[1] = h.y;
            }
        }
    } else {  // (an odd D ends on a single element)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const bool ok = k + e < A.D;
            c.p[e] = ok ? p[e] : 0.f; c.m[e] = ok ? m[e] : 0.f; c.v[e] = ok ? v[e] : 0.f;
            if (g && ok) c.g[e] = g[e];
        }
    }
}

// consts[s] for a wave-uniform s, read through the SCALAR cache (constant address space: s_load_dwordx2).  As an ordinary global
// read the compiler has to assume the table changes under the launch (block 0 records consts[A.step], which is never read
// back here: replay_consts) and issues a vector load plus s_waitcnt vmcnt(0) per replayed step -- an L1 / L2 round trip in every
// link of the serial chain (round 5, ISA of the unroll-1 loop: two of them per step).
__device__ __forceinline__ float2 consts_at(const AdamRowArgs &A, int s) {
    typedef const float __attribute__((address_space(4))) *cptr;
    cptr t = (cptr)(uintptr_t)(A.consts + s);
    return make_float2(t[0], t[1]);
}

__device__ __forceinline__ float2 replay_consts(const AdamRowArgs &A, int s) {
    // advance form: consts[A.step] is being written by this very launch -- take it from the arguments
    return (A.g && s == A.step) ? make_float2(A.neg_step, A.sqrt_bc2) : consts_at(A, s);
}

// replay the steps (from, to] of one chunk: the first with the row's gradient in the advance form, the rest with a zero
// gradient.  A serial chain per element (2 transcendentals per step); a row's gap can be hundreds of steps (entities only
// the random pool ever touches), so the chain is kept short: half a workgroup (512 lanes x float4) or a whole one
// (1024 lanes x float2) per row.
template <int EPL, int UNROLL = kReplayUnroll>
__device__ __forceinline__ void replay_finish(const AdamRowArgs &A, int64_t row, int k, int from, int to, RowChunk<EPL> &c) {
    f2 p[EPL / 2], m[EPL / 2], v[EPL / 2];
#pragma unroll
    for (int e = 0; e < EPL / 2; ++e) {
        p[e] = f2{c.p[2 * e], c.p[2 * e + 1]}; m[e] = f2{c.m[2 * e], c.m[2 * e + 1]}; v[e] = f2{c.v[2 * e], c.v[2 * e + 1]};
    }
    int s = from + 1;
    bool clear = false;
#if defined(MKB_ADAM_MEASURE) && (MKB_ADAM_MEASURE & 1)  // (measurement builds, WRONG results: rows move, nothing is replayed)
    s = to + 1;
    if (false)
#else
    if (A.g)
#endif
    {  // the row's first pending step is the one its gradient row belongs to
        const float2 cs = replay_consts(A, s);
        const float inv = __builtin_amdgcn_rcpf(cs.y);
#pragma unroll
        for (int e = 0; e < EPL / 2; ++e) {
            const f2 g = f2{c.g[2 * e], c.g[2 * e + 1]};
            clear |= g.x != 0.f || g.y != 0.f;
            adam_pair(p[e], g, m[e], v[e], A.w1, A.b2, A.w2, cs.x, inv, A.eps);
        }
        ++s;
    }
    // Zero-gradient steps.  `from` / `to` are wave-uniform (scalar registers), so the per-step constants come through the
    // scalar cache; they are fetched UNROLL steps ahead of their use, and the steps of a group are written out side by
    // side: per element the only serial links between steps are ONE multiply (v), ONE fma (m) and ONE add (p) -- the
    // sqrt / rcp chains of neighbouring steps overlap.  (Before: one vector load round trip + a 7-deep chain per step; the
    // launch lasted as long as the row with the longest gap -- WN18RR: ~900 pending steps.)
    const int last_tab = A.g ? to - 1 : to;  // the advance form's own step is recorded by this very launch: not in the table
    if (s + UNROLL - 1 <= last_tab) {
        float2 nx[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) nx[u] = consts_at(A, s + u);
        for (;;) {
            float2 cs[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) cs[u] = nx[u];
            s += UNROLL;
            const bool more = s + UNROLL - 1 <= last_tab;
            if (more) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) nx[u] = consts_at(A, s + u);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const float inv = __builtin_amdgcn_rcpf(cs[u].y);
#pragma unroll
                for (int e = 0; e < EPL / 2; ++e) adam_pair_zero_grad(p[e], m[e], v[e], A.w1, A.b2, cs[u].x, inv, A.eps);
            }
            if (!more) break;
        }
    }
    for (; s <= to; ++s) {
        const float2 cs = replay_consts(A, s);
        const float inv = __builtin_amdgcn_rcpf(cs.y);
#pragma unroll
        for (int e = 0; e < EPL / 2; ++e) adam_pair_zero_grad(p[e], m[e], v[e], A.w1, A.b2, cs.x, inv, A.eps);
    }
    float *pp = A.p + row * A.D + k, *mm = A.m + row * A.D + k, *vv = A.v + row * A.D + k;
    float *gg = A.g ? A.g + row * A.D + k : nullptr;
    if (k + EPL <= A.D && (A.D % EPL) == 0) {
        if constexpr (EPL == 4) {
            *reinterpret_cast<float4 *>(pp) = make_float4(p[0].x, p[0].y, p[1].x, p[1].y);
            *reinterpret_cast<float4 *>(mm) = make_float4(m[0].x, m[0].y, m[1].x, m[1].y);
            *reinterpret_cast<float4 *>(vv) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
            if (clear) *reinterpret_cast<float4 *>(gg) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            *reinterpret_cast<float2 *>(pp) = make_float2(p[0].x, p[0].y);
            *reinterpret_cast<float2 *>(mm) = make_float2(m[0].x, m[0].y);
            *reinterpret_cast<float2 *>(vv) = make_float2(v[0].x, v[0].y);
            if (clear) *reinterpret_cast<float2 *>(gg) = make_float2(0.f, 0.f);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (k + e < A.D) {
                pp[e] = p[e / 2][e % 2]; mm[e] = m[e / 2][e % 2]; vv[e] = v[e / 2][e % 2];
                if (clear) gg[e] = 0.f;
            }
    }
}

// one row: LANES lanes x EPL elements per pass.  The first chunk is requested BEFORE the ownership exchange returns (a
// global atomic round trip ahead of the row's HBM latency otherwise); lanes that turn out to have nothing to do drop it.
template <int EPL, int UNROLL>
__device__ __forceinline__ void replay_row_block(const AdamRowArgs &A, int64_t row, bool valid, int lane, int lanes, int *s_old,
                                                 bool ahead) {
    // whole waves per row: the row index is wave-uniform -- say so, and the four row addresses become one scalar base each plus a
    // 32-bit lane offset instead of four 64-bit vector addresses (registers: the point is a fourth workgroup per CU)
    row = ((int64_t)__builtin_amdgcn_readfirstlane((int)(row >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)row);
    valid = __builtin_amdgcn_readfirstlane((int)valid) != 0;
#if defined(MKB_ADAM_MEASURE) && (MKB_ADAM_MEASURE & 2)  // (measurement builds, WRONG results: no ownership exchange)
    if (lane == 0 && valid) *s_old = A.step - 2;
#else
    if (lane == 0 && valid) *s_old = atomicExch(&A.last[row], A.step);  // first claimant of a duplicated id does the work
#endif
    const int k0 = lane * EPL, D = (int)A.D;  // (row lengths fit 31 bits: fill_args)
    RowChunk<EPL> c;
    if (valid && ahead && k0 < D) replay_load<EPL>(A, row, k0, c);
    __syncthreads();
    if (!valid) return;
    const int old = __builtin_amdgcn_readfirstlane(*s_old);  // (whole waves per row: uniform)
    if (old <= 0 || old >= A.step) return;  // never touched (m = v = 0: identity) or already current
    for (int k = k0; k < D; k += EPL * lanes) {
        if (!ahead || k != k0) replay_load<EPL>(A, row, k, c);
        replay_finish<EPL, UNROLL>(A, row, k, old, A.step, c);
    }
}

// UNROLL: pending zero-gradient steps replayed side by side.  4 (kReplayUnroll) where rows wait many steps between visits (the
// chains are serial: WN18RR ~16 steps on average, YAGO3-10 up to the sweep period); 1 where they wait a few (FB15k-237: 5.7):
// the replay is then a small part of the block's life, and 70 instead of 106 VGPRs let a third block share the CU (round 5:
// 39.6 -> 37.6 us for the headline's launch; squeezed to 63 VGPRs for a fourth block it spills: 49 us).  The arithmetic per element and step is the same function either way: same bits.
template <int UNROLL>
__global__ __launch_bounds__(kCatchThreads) void adam_rows_catchup_kernel(AdamRowArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lds_draw[];  // only sized when a draw block rides
    if (A.g && A.step > 0 && blockIdx.x == 0 && threadIdx.x == 0)
        A.consts[A.step] = make_float2(A.neg_step, A.sqrt_bc2);  // advance form: the launch after a deferred step records it
    if (A.first_row_block && blockIdx.x == 0) {  // dispatched first: ~8 us of serial work in the shadow of the row blocks
        pool_draw_body<kCatchThreads>(A.draw, lds_draw);
        return;
    }
    if ((int)blockIdx.x < A.first_row_block + A.n_filter) {
        filter_rows_body<kCatchThreads>(A.filt, (int)blockIdx.x - A.first_row_block, reinterpret_cast<int32_t *>(lds_draw));
        return;
    }
    const int64_t bid = (int64_t)blockIdx.x - A.first_row_block - A.n_filter;
    if (bid >= A.n_ids) {  // the dense rider of a deferred step: the last blocks
        const int64_t nb = (int64_t)gridDim.x - A.first_row_block - A.n_filter - A.n_ids;
        adam_dense_range(A.dp, A.dg, A.dm, A.dv, A.dn, (bid - A.n_ids) * kCatchThreads + threadIdx.x, nb * kCatchThreads, A.w1,
                         A.b2, A.w2, A.d_neg_step, A.d_sqrt_bc2, A.eps, 1);
        return;
    }
    __shared__ int s_old2[16];
    const bool ahead = A.ids || A.seg_pool || A.own_ids;  // (a flush walks every row, most of them with nothing pending: no guessing there)
    const int rpb = A.rows_per_block, lanes = kCatchThreads / rpb;
    const int sub = (int)threadIdx.x / lanes, lane = (int)threadIdx.x - sub * lanes;
    const int64_t r = bid * rpb + sub;
    bool valid = r < A.n_rows_listed;
    int64_t row = 0;
    if (valid) {
        if (r >= A.n_batch_rows) {
            row = (int64_t)A.sweep_start + (r - A.n_batch_rows);
            if (row >= A.n_table) row -= A.n_table;
        } else if (A.own_ids && r < A.own_n) {
            const int64_t e = A.own_ids[r];
            row = (int)(e % A.own_world) == A.own_rank ? e / A.own_world : -1;
        } else if (A.own_ids) row = A.ids[r - A.own_n];
        else if (A.ids) row = A.ids[r];
        else if (A.seg_pool) row = r < A.seg_P ? A.seg_pool[r]
                                 : (r < A.seg_P + A.seg_B ? A.seg_sample[3 * (r - A.seg_P)]
                                                          : A.seg_sample[3 * (r - A.seg_P - A.seg_B) + 2]);
        else row = r;
    }
    // a negative id = "not a row of this table" (an entry another rank owns): skipped.  So is an id past the table (n_table is set
    // whenever the rows come from an id list): the caller's id check raises the reference's IndexError for it afterwards
    // (mkb_check_ids sets its flag asynchronously) -- the replay must not have written p / m / v / last out of bounds by then
    if (row < 0 || (A.n_table > 0 && row >= A.n_table)) { valid = false; row = 0; }
    if (A.vec4) replay_row_block<4, UNROLL>(A, row, valid, lane, lanes, &s_old2[sub], ahead);
    else replay_row_block<2, UNROLL>(A, row, valid, lane, lanes, &s_old2[sub], ahead);
}

// A flush (every row of the table, no id list, no riders) is a stream over the whole table: p, m, v of every row with pending
// steps in and out (FB15k-237 / hidden 1000: 700 MB).  Through the catch-up kernel above -- 107 VGPRs for its four steps replayed
// side by side, one 1024-lane workgroup per CU -- it runs at 3.4 TB/s (0.22 ms: 11 us per step of a 20-step run).  Here: one
// 256-lane workgroup per row, the steps replayed ONE at a time (same functions, same order of operations per element: the same bits),
// few registers, eight waves per SIMD to keep HBM busy.
__global__ __launch_bounds__(256) void adam_rows_flush_kernel(AdamRowArgs A) {
    __shared__ int s_old;
    const int64_t row = blockIdx.x;
    if (A.g && A.step > 0 && blockIdx.x == 0 && threadIdx.x == 0) A.consts[A.step] = make_float2(A.neg_step, A.sqrt_bc2);
    if (row >= A.n_rows_listed) {  // the dense rider of a deferred step (the small relation table): the last blocks
        const int64_t nb = (int64_t)gridDim.x - A.n_rows_listed;
        adam_dense_range(A.dp, A.dg, A.dm, A.dv, A.dn, (row - A.n_rows_listed) * 256 + threadIdx.x, nb * 256, A.w1, A.b2, A.w2,
                         A.d_neg_step, A.d_sqrt_bc2, A.eps, 1);
        return;
    }
    if (threadIdx.x == 0) s_old = atomicExch(&A.last[row], A.step);
    __syncthreads();
    const int old = __builtin_amdgcn_readfirstlane(s_old);
    if (old <= 0 || old >= A.step) return;  // never touched (m = v = 0: identity) or already current
    if (A.vec4) {
        for (int64_t k = (int64_t)threadIdx.x * 4; k < A.D; k += 1024) {
            RowChunk<4> c;
            replay_load<4>(A, row, k, c);
            replay_finish<4, 1>(A, row, k, old, A.step, c);
        }
    } else {
        for (int64_t k = (int64_t)threadIdx.x * 2; k < A.D; k += 512) {
            RowChunk<2> c;
            replay_load<2>(A, row, k, c);
            replay_finish<2, 1>(A, row, k, old, A.step, c);
        }
    }
}

__global__ __launch_bounds__(256) void adam_rows_step_kernel(AdamRowArgs A) {
    __shared__ int s_old;
    const int bid = (int)blockIdx.x;
    if (bid >= A.n_ids) {  // the dense rider: the last blocks
        const int64_t nb = (int64_t)gridDim.x - A.n_ids;
        adam_dense_range(A.dp, A.dg, A.dm, A.dv, A.dn, ((int64_t)bid - A.n_ids) * 256 + threadIdx.x, nb * 256, A.w1,
                         A.b2, A.w2, A.d_neg_step, A.d_sqrt_bc2, A.eps, 1);
        return;
    }
    if (bid == 0 && threadIdx.x == 0) A.consts[A.step] = make_float2(A.neg_step, A.sqrt_bc2);
    const int64_t row = A.ids[bid];
    if (row < 0 || (A.n_table > 0 && row >= A.n_table)) return;  // (workgroup-uniform) an entry another rank owns in a row-sharded table's id list, or an id past the table
    if (threadIdx.x == 0) s_old = atomicExch(&A.last[row], A.step);
    float *p = A.p + row * A.D, *g = A.g + row * A.D, *m = A.m + row * A.D, *v = A.v + row * A.D;
    if ((A.D & 3) == 0) {  // 16-byte aligned rows: one float4 per lane and array
        float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g);
        float4 *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
        // the first float4s are requested BEFORE the ownership exchange returns (a global atomic round trip): a duplicate's
        // workgroup wastes four cached loads, every owner saves that latency
        const int64_t k0 = threadIdx.x, n4 = A.D >> 2;
        float4 pp0, gg0, mm0, vv0;
        if (k0 < n4) { pp0 = p4[k0]; gg0 = g4[k0]; mm0 = m4[k0]; vv0 = v4[k0]; }
        __syncthreads();
        if (s_old == A.step) return;  // duplicate id: another workgroup owns this row
        for (int64_t k = k0; k < n4; k += 256) {
            float4 pp, gg, mm, vv;
            if (k == k0) { pp = pp0; gg = gg0; mm = mm0; vv = vv0; }
            else { pp = p4[k]; gg = g4[k]; mm = m4[k]; vv = v4[k]; }
            adam_one(pp.x, gg.x, mm.x, vv.x, A.w1, A.b2, A.w2, A.neg_step, A.sqrt_bc2, A.eps);
            adam_one(pp.y, gg.y, mm.y, vv.y, A.w1, A.b2, A.w2, A.neg_step, A.sqrt_bc2, A.eps);
            adam_one(pp.z, gg.z, mm.z, vv.z, A.w1, A.b2, A.w2, A.neg_step, A.sqrt_bc2, A.eps);
            adam_one(pp.w, gg.w, mm.w, vv.w, A.w1, A.b2, A.w2, A.neg_step, A.sqrt_bc2, A.eps);
            p4[k] = pp; m4[k] = mm; v4[k] = vv; g4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    __syncthreads();
    if (s_old == A.step) return;
    for (int64_t k = threadIdx.x; k < A.D; k += 256) {
        float pp = p[k], mm = m[k], vv = v[k];
        adam_one(pp, g[k], mm, vv, A.w1, A.b2, A.w2, A.neg_step, A.sqrt_bc2, A.eps);
        p[k] = pp; m[k] = mm; v[k] = vv; g[k] = 0.f;
    }
}

static int fill_args(AdamRowArgs &A, float *param, float *grad, float *m, float *v, int32_t *last, float *consts,
                     const int64_t *ids, int64_t D, int64_t step, float lr, float beta1, float beta2, float eps) {
    MKB_REQUIRE(param && m && v && last && consts, "null pointer");
    MKB_REQUIRE(D > 0 && D < INT32_MAX && step >= 0 && step < INT32_MAX, "bad D / step");
    A.p = param; A.g = grad; A.m = m; A.v = v; A.last = last; A.consts = (float2 *)consts; A.ids = ids; A.D = D;
    A.step = (int32_t)step;
    A.w1 = (float)(1.0 - (double)beta1); A.b2 = beta2; A.w2 = (float)(1.0 - (double)beta2); A.eps = eps;
    const double s = step > 0 ? (double)step : 1.0;
    A.neg_step = (float)(-((double)lr / (1.0 - pow((double)beta1, s))));
    A.sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, s));
    return MKB_OK;
}

}  // namespace mkb

namespace mkb {

// Sweep.  Exact dense Adam moves every parameter every step, so the replay's arithmetic is N x D element-steps per step no
// matter when it happens (measured: ~85 ns per wave and pending step at 4 elements per lane: WN18RR +16 us, FB15k-237
// +12 us, YAGO3-10 ~40 us per launch if spread evenly).  It is NOT spread evenly when entities are rare: a YAGO3-10 entity
// that only the random pool reaches waits ~1,500 steps in the tail, its chain is serial, and the launch lasts as long as
// its longest chain (99 us measured).  So when the mean gap N / rows-per-step is large, each per-step launch also visits
// a moving window of min(N / period, rows-of-this-launch) rows: every row is brought up to date at least once per
// max(period, N / rows-per-launch) steps (the window never exceeds the launch's own row count, so for N > period * rows the
// bound is N / rows; the window's start is a function of the step and of THAT launch's window, so launches with different row
// counts -- a catch-up of a few ids between two training steps -- move it unevenly: a latency aid, not a guarantee).  Same
// arithmetic, same results bit for bit (every pending step of every row is replayed exactly once, whenever that happens);
// cost: the window's rows x (p, m, v) in + out.  YAGO3-10: 99 -> 64 us per launch, step 0.268 -> 0.211 ms; WN18RR / FB15k-237
// (mean gap 18 / 6 steps, every entity is a positive often enough): no gain, so no sweep.
// Round 5, with the replay's per-step round trips gone and in runs long enough for the gaps to reach their steady state (1,500
// steps): WN18RR's launch 36.5 us without a sweep, 31.3-31.9 with a period of 24-48 (step 0.1193 -> 0.1144 ms), 32.5 at 128;
// YAGO3-10 91.8 without, 53.1 / 54.5 / 65.1 / 79.0 at 32 / 64 / 128 / 256; FB15k-237 37.7 without, 37.7-39.1 with.  Hence: a
// period of 32 from a mean gap of 12 steps on.
constexpr int kSweepPeriod = 32, kSweepMinGap = 12;

static void set_row_blocks(AdamRowArgs &A, int64_t rows, int64_t n_table = 0) {
    int64_t sweep = 0;
    A.n_batch_rows = (int32_t)rows; A.sweep_start = 0; A.n_table = n_table;
    if (rows > 0 && n_table > 0 && A.step > 0) {
        const char *fe = getenv("MKB_ADAM_SWEEP");  // period, 0 = off (read per call: the tests switch it within one process)
        const int forced = fe ? atoi(fe) : -1;
        if (forced > 0) sweep = (n_table + forced - 1) / forced;
        else if (forced < 0 && n_table >= (int64_t)kSweepMinGap * rows) sweep = std::min((n_table + kSweepPeriod - 1) / kSweepPeriod, rows);
        if (sweep > 0) {
            A.sweep_start = (int32_t)((((int64_t)A.step - 1) * sweep) % n_table);
            rows += sweep;
        }
    }
    bool vec4 = (A.D & 3) == 0 && (((uintptr_t)A.p | (uintptr_t)A.m | (uintptr_t)A.v | (uintptr_t)A.g) & 15) == 0;
    A.vec4 = vec4 ? 1 : 0;
    // as many rows per workgroup (kCatchThreads lanes) as fit with one chunk per lane (whole waves per row): at 1024 lanes 2000-float rows 2,
    // 1000-float rows 4, the 250-float rows of an 8-way dimension shard 8 -- short rows used to idle most of the lanes
    const int64_t per_lane = vec4 ? 4 : 2;
    int64_t lanes = ((A.D + per_lane - 1) / per_lane + 63) / 64 * 64;
    int rpb = 1;
    while (rpb < 16 && (int64_t)kCatchThreads / (rpb * 2) >= lanes) rpb *= 2;
    A.rows_per_block = rpb;
    A.n_rows_listed = (int32_t)rows;
    A.n_ids = (int32_t)((rows + A.rows_per_block - 1) / A.rows_per_block);
}

// dense rider of an advance launch -> number of extra workgroups (0 = none)
static int attach_rider(AdamRowArgs &A, const mkb_adam_dense_t *rider, float lr, float beta1, float beta2, int threads,
                        int64_t *extra) {
    *extra = 0;
    if (!rider || rider->n <= 0) return MKB_OK;
    MKB_REQUIRE(rider->param && rider->grad && rider->exp_avg && rider->exp_avg_sq && rider->step >= 1, "bad dense rider");
    MKB_REQUIRE((((uintptr_t)rider->param | (uintptr_t)rider->grad | (uintptr_t)rider->exp_avg | (uintptr_t)rider->exp_avg_sq) & 15) == 0,
                "buffers must be 16-byte aligned");
    A.dp = rider->param; A.dg = rider->grad; A.dm = rider->exp_avg; A.dv = rider->exp_avg_sq; A.dn = rider->n;
    A.d_neg_step = (float)(-((double)lr / (1.0 - pow((double)beta1, (double)rider->step))));
    A.d_sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)rider->step));
    int64_t e = ((rider->n >> 2) + threads - 1) / threads;
    *extra = e < 1 ? 1 : (e > 1024 ? 1024 : e);
    return MKB_OK;
}

// mean gap between two visits of a row = table rows / rows a launch lists: short gaps take the lean replay (see the kernel)
static bool short_gaps(const AdamRowArgs &A) {
    if (const char *e = getenv("MKB_ADAM_UNROLL")) return atoi(e) == 1;  // A/B switch (read per call)
    return A.n_table > 0 && A.n_batch_rows > 0 && (int64_t)A.n_table < (int64_t)12 * A.n_batch_rows;
}

static int rows_advance(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                        int64_t n_rows, int64_t D, const int64_t *ids, int64_t n_ids, int64_t step_upto, float lr, float beta1,
                        float beta2, float eps, const mkb_adam_dense_t *rider, mkb_sampler_t *draw_ahead, void *stream,
                        const int64_t *own_ids = nullptr, int64_t own_n = 0, int own_world = 0, int own_rank = 0) {
    AdamRowArgs A{};
    if (int rc = fill_args(A, param, grad, exp_avg, exp_avg_sq, last, consts, ids, D, step_upto, lr, beta1, beta2, eps)) return rc;
    int64_t n = ids ? n_ids : n_rows;
    if (own_ids) {
        MKB_REQUIRE(own_n > 0 && own_n <= INT32_MAX && own_world >= 1 && own_rank >= 0 && own_rank < own_world, "bad ownership");
        MKB_REQUIRE(ids || n_ids == 0, "null local id list");
        A.own_ids = own_ids; A.own_n = (int32_t)own_n; A.own_world = own_world; A.own_rank = own_rank;
        n = own_n + n_ids;
    }
    if (n <= 0 || step_upto <= 0) n = 0;  // nothing can be pending before the first step
    MKB_REQUIRE(n <= INT32_MAX, "too many rows");
    set_row_blocks(A, n, (ids || own_ids) ? n_rows : 0);  // (a flush walks the whole table anyway)
    n = A.n_ids;
    int64_t extra = 0;
    static const bool no_flush_kernel = getenv("MKB_ADAM_NO_FLUSH_KERNEL") != nullptr;  // A/B switch
    if (!ids && !own_ids && A.n_rows_listed > 0 && A.n_rows_listed == A.n_batch_rows && !no_flush_kernel) {  // the whole table
        if (int rc = attach_rider(A, rider, lr, beta1, beta2, 256, &extra)) return rc;
        ProfScope ps(MKB_PROF_ADAM, (hipStream_t)stream);
        hipLaunchKernelGGL(adam_rows_flush_kernel, dim3((unsigned)(A.n_rows_listed + extra)), dim3(256), 0, (hipStream_t)stream, A);
        MKB_LAUNCH_CHECK();
        return MKB_OK;
    }
    if (int rc = attach_rider(A, rider, lr, beta1, beta2, kCatchThreads, &extra)) return rc;
    if (n + extra == 0) return MKB_OK;
    size_t lds = 0;
    if (draw_ahead && (ids || own_ids) && sampler_draw_ahead(draw_ahead, &A.draw, &lds)) A.first_row_block = 1;
    ProfScope ps(MKB_PROF_ADAM, (hipStream_t)stream);
    if (short_gaps(A))
        hipLaunchKernelGGL(adam_rows_catchup_kernel<1>, dim3((unsigned)(n + extra + A.first_row_block)), dim3(kCatchThreads), lds,
                           (hipStream_t)stream, A);
    else
        hipLaunchKernelGGL(adam_rows_catchup_kernel<kReplayUnroll>, dim3((unsigned)(n + extra + A.first_row_block)), dim3(kCatchThreads),
                           lds, (hipStream_t)stream, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

static int rows_advance_generate(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                 int64_t n_rows, int64_t D, int64_t step_upto, float lr, float beta1, float beta2, float eps,
                                 const mkb_adam_dense_t *rider, mkb_sampler_t *sampler, const int64_t *sample, int64_t B,
                                 int mode, int64_t *neg, int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched,
                                 void *stream, int own_world = 0, int own_rank = 0, const int64_t *local_ids = nullptr,
                                 int64_t n_local_ids = 0) {
    hipStream_t st = (hipStream_t)stream;
    AdamRowArgs A{};
    if (int rc = fill_args(A, param, grad, exp_avg, exp_avg_sq, last, consts, nullptr, D, step_upto > 0 ? step_upto : 0, lr,
                           beta1, beta2, eps)) return rc;
    size_t lds = 0;
    ProfScope ps(MKB_PROF_SAMPLER, st);
    if (int rc = sampler_ride(sampler, sample, B, mode, neg, pool, pos, cnt, touched, &A.filt, &A.draw, &A.seg_pool, &lds, st, kCatchThreads))
        return rc;
    A.first_row_block = 1;
    A.n_filter = (int32_t)((B + A.filt.rows_per_wg - 1) / A.filt.rows_per_wg);  // one wave per row, <= kCatchThreads / 64 rows per workgroup
    A.seg_sample = sample; A.seg_P = A.filt.P; A.seg_B = (int32_t)B;
    if (own_world > 0) {
        // shard of a row-sharded table: the rows to visit are the pool ids this rank owns (the sampler's pool buffer holds
        // GLOBAL ids) followed by the shard indices other ranks asked for; the batch's own heads / tails live elsewhere
        MKB_REQUIRE(own_rank >= 0 && own_rank < own_world && (local_ids || n_local_ids == 0) && n_local_ids >= 0, "bad ownership");
        A.own_ids = A.seg_pool; A.own_n = A.filt.P; A.own_world = own_world; A.own_rank = own_rank;
        A.ids = local_ids; A.seg_pool = nullptr;
        set_row_blocks(A, step_upto > 0 ? (int64_t)A.own_n + n_local_ids : 0, n_rows);
    } else
    set_row_blocks(A, step_upto > 0 ? (int64_t)A.seg_P + 2 * B : 0, n_rows);  // nothing is pending before the first step
    const int64_t rows = A.n_ids;
    int64_t extra = 0;
    if (int rc = attach_rider(A, rider, lr, beta1, beta2, kCatchThreads, &extra)) return rc;
    static LdsOptIn big_lds[2];  // 16 rows per filter workgroup need ~73 KB of dynamic LDS: opt in once per device (160 KB per CU)
    const dim3 grid((unsigned)(1 + A.n_filter + rows + extra));
    if (short_gaps(A)) {
        if (int rc = big_lds[0].ensure(reinterpret_cast<const void *>(&adam_rows_catchup_kernel<1>), 96 * 1024)) return rc;
        hipLaunchKernelGGL(adam_rows_catchup_kernel<1>, grid, dim3(kCatchThreads), lds, st, A);
    } else {
        if (int rc = big_lds[1].ensure(reinterpret_cast<const void *>(&adam_rows_catchup_kernel<kReplayUnroll>), 96 * 1024)) return rc;
        hipLaunchKernelGGL(adam_rows_catchup_kernel<kReplayUnroll>, grid, dim3(kCatchThreads), lds, st, A);
    }
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

}  // namespace mkb

extern "C" int mkb_adam_rows_catchup(float *param, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                     int64_t n_rows, int64_t D, const int64_t *ids, int64_t n_ids, int64_t step_upto,
                                     float beta1, float beta2, float eps, mkb_sampler_t *draw_ahead, void *stream) {
    return mkb::rows_advance(param, nullptr, exp_avg, exp_avg_sq, last, consts, n_rows, D, ids, n_ids, step_upto, 0.f, beta1,
                             beta2, eps, nullptr, draw_ahead, stream);
}

// mkb_sampler_generate and mkb_adam_rows_catchup(ids = the batch's pool | heads | tails) as ONE launch, plus the draw of
// the next pool: block 0 draws, the next ceil(B / 16) blocks filter this batch's rows, the rest replay the pending steps.
extern "C" int mkb_adam_rows_catchup_generate(float *param, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                              int64_t n_rows, int64_t D, int64_t step_upto, float beta1, float beta2, float eps,
                                              mkb_sampler_t *sampler, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                                              int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream) {
    return mkb::rows_advance_generate(param, nullptr, exp_avg, exp_avg_sq, last, consts, n_rows, D, step_upto, 0.f, beta1, beta2, eps,
                                      nullptr, sampler, sample, B, mode, neg, pool, pos, cnt, touched, stream);
}

// Advance form of the two calls above (see the top of the row-lazy section): `grad` = the table's dense gradient, whose
// rows the replay consumes (first pending step of each row) and clears; `lr` = learning rate of step `step_upto` (the
// step whose own launch was skipped); rider = the small dense tensor of that step, or null.
extern "C" int mkb_adam_rows_advance(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                     int64_t n_rows, int64_t D, const int64_t *ids, int64_t n_ids, int64_t step_upto, float lr,
                                     float beta1, float beta2, float eps, const mkb_adam_dense_t *rider,
                                     mkb_sampler_t *draw_ahead, void *stream) {
    MKB_REQUIRE(grad, "null gradient (use mkb_adam_rows_catchup)");
    return mkb::rows_advance(param, grad, exp_avg, exp_avg_sq, last, consts, n_rows, D, ids, n_ids, step_upto, lr, beta1, beta2,
                             eps, rider, draw_ahead, stream);
}

// Row-sharded table: the rows to visit are given as GLOBAL entity ids (the candidate pool, the same on every rank: entries
// another rank owns are skipped) followed by shard indices (the rows other ranks asked this owner for).  grad == null: the
// plain catch-up (no deferred real step).
extern "C" int mkb_adam_rows_advance_sharded(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last,
                                             float *consts, int64_t n_rows, int64_t D, const int64_t *global_ids,
                                             int64_t n_global, int world, int rank, const int64_t *local_ids,
                                             int64_t n_local_ids, int64_t step_upto, float lr, float beta1, float beta2,
                                             float eps, const mkb_adam_dense_t *rider, mkb_sampler_t *draw_ahead, void *stream) {
    MKB_REQUIRE(global_ids && n_global > 0, "null global id list (use mkb_adam_rows_advance)");
    MKB_REQUIRE(grad || !rider, "a dense rider needs the advance form (grad != null)");
    return mkb::rows_advance(param, grad, exp_avg, exp_avg_sq, last, consts, n_rows, D, local_ids, n_local_ids, step_upto,
                             grad ? lr : 0.f, beta1, beta2, eps, rider, draw_ahead, stream, global_ids, n_global, world, rank);
}

extern "C" int mkb_adam_rows_advance_generate(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last,
                                              float *consts, int64_t n_rows, int64_t D, int64_t step_upto, float lr, float beta1,
                                              float beta2, float eps, const mkb_adam_dense_t *rider, mkb_sampler_t *sampler,
                                              const int64_t *sample, int64_t B, int mode, int64_t *neg, int64_t *pool,
                                              int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream) {
    MKB_REQUIRE(grad, "null gradient (use mkb_adam_rows_catchup_generate)");
    return mkb::rows_advance_generate(param, grad, exp_avg, exp_avg_sq, last, consts, n_rows, D, step_upto, lr, beta1, beta2, eps, rider,
                                      sampler, sample, B, mode, neg, pool, pos, cnt, touched, stream);
}

// mkb_adam_rows_advance_sharded with this rank's mkb_sampler_generate (filter of its rows + the next pool's draw) in the same
// launch: global_ids are the sampler's own pool.  grad == null: plain catch-up.
extern "C" int mkb_adam_rows_advance_sharded_generate(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last,
                                                      float *consts, int64_t n_rows, int64_t D, int world, int rank,
                                                      const int64_t *local_ids, int64_t n_local_ids, int64_t step_upto, float lr,
                                                      float beta1, float beta2, float eps, const mkb_adam_dense_t *rider,
                                                      mkb_sampler_t *sampler, const int64_t *sample, int64_t B, int mode,
                                                      int64_t *neg, int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched,
                                                      void *stream) {
    MKB_REQUIRE(world >= 1, "bad world");
    MKB_REQUIRE(grad || !rider, "a dense rider needs the advance form (grad != null)");
    return mkb::rows_advance_generate(param, grad, exp_avg, exp_avg_sq, last, consts, n_rows, D, step_upto, grad ? lr : 0.f, beta1, beta2,
                                      eps, rider, sampler, sample, B, mode, neg, pool, pos, cnt, touched, stream, world, rank,
                                      local_ids, n_local_ids);
}

extern "C" int mkb_adam_rows_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                  int64_t n_rows, int64_t D, const int64_t *ids, int64_t n_ids, int64_t step, float lr,
                                  float beta1, float beta2, float eps, const mkb_adam_dense_t *rider, void *stream) {
    mkb::AdamRowArgs A{};
    if (int rc = mkb::fill_args(A, param, grad, exp_avg, exp_avg_sq, last, consts, ids, D, step, lr, beta1, beta2, eps)) return rc;
    MKB_REQUIRE(grad && ids && step >= 1 && n_ids > 0 && n_ids <= INT32_MAX, "bad arguments");
    A.n_ids = (int32_t)n_ids;
    A.n_table = n_rows;  // (ids past the table are skipped, not stepped out of bounds)
    int64_t extra = 0;
    if (rider && rider->n > 0) {
        MKB_REQUIRE(rider->param && rider->grad && rider->exp_avg && rider->exp_avg_sq && rider->step >= 1, "bad dense rider");
        MKB_REQUIRE((((uintptr_t)rider->param | (uintptr_t)rider->grad | (uintptr_t)rider->exp_avg | (uintptr_t)rider->exp_avg_sq) & 15) == 0,
                    "buffers must be 16-byte aligned");
        A.dp = rider->param; A.dg = rider->grad; A.dm = rider->exp_avg; A.dv = rider->exp_avg_sq; A.dn = rider->n;
        A.d_neg_step = (float)(-((double)lr / (1.0 - pow((double)beta1, (double)rider->step))));
        A.d_sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)rider->step));
        extra = ((rider->n >> 2) + 255) / 256;
        extra = extra < 1 ? 1 : (extra > 1024 ? 1024 : extra);
    }
    mkb::ProfScope ps(MKB_PROF_ADAM, (hipStream_t)stream);
    hipLaunchKernelGGL(mkb::adam_rows_step_kernel, dim3((unsigned)(n_ids + extra)), dim3(256), 0, (hipStream_t)stream, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

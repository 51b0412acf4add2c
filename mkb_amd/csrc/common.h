// Host-side helpers shared by the C-ABI entry points of libmkb_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mkb_hip.h"

namespace mkb {

int set_error(int code, const char *fmt, ...);

#define MKB_CHECK_HIP(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::mkb::set_error(MKB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                    __FILE__, __LINE__);                                        \
    } while (0)

#define MKB_REQUIRE(cond, ...)                                          \
    do {                                                                \
        if (!(cond)) return ::mkb::set_error(MKB_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define MKB_LAUNCH_CHECK() MKB_CHECK_HIP(hipGetLastError())

// Opt-in to more than 64 KB of dynamic LDS for a kernel.  The attribute is per DEVICE (and per kernel), so what has been
// granted is remembered per device: a second GPU driven from the same process gets its own opt-in.  One object per kernel
// (instantiation): `static LdsOptIn grant; if (int rc = grant.ensure(fn, bytes)) return rc;`
struct LdsOptIn {
    static constexpr int kMaxDevices = 64;
    size_t granted[kMaxDevices] = {};
    int ensure(const void *fn, size_t bytes) {
        if (bytes <= 64 * 1024) return MKB_OK;
        int dev = 0;
        MKB_CHECK_HIP(hipGetDevice(&dev));
        const bool tracked = dev >= 0 && dev < kMaxDevices;
        if (tracked && bytes <= granted[dev]) return MKB_OK;
        MKB_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (tracked) granted[dev] = bytes;
        return MKB_OK;
    }
};

inline int validate_tables(const mkb_tables_t *tb) {
    MKB_REQUIRE(tb != nullptr, "tables is null");
    MKB_REQUIRE(tb->model >= MKB_TRANSE && tb->model <= MKB_PROTATE, "unknown model id %d", tb->model);
    MKB_REQUIRE(tb->ent && tb->rel, "null embedding table");
    MKB_REQUIRE(tb->n_entity > 0 && tb->n_relation > 0 && tb->hidden_dim > 0, "empty table");
    const int64_t d = tb->hidden_dim;
    int64_t de = d, dr = d;
    if (tb->model == MKB_ROTATE) de = 2 * d;
    if (tb->model == MKB_COMPLEX) { de = 2 * d; dr = 2 * d; }
    MKB_REQUIRE(tb->entity_dim == de && tb->relation_dim == dr,
                "entity_dim/relation_dim (%lld, %lld) do not match model %d with hidden_dim %lld",
                (long long)tb->entity_dim, (long long)tb->relation_dim, tb->model, (long long)d);
    MKB_REQUIRE(tb->model != MKB_PROTATE || tb->modulus != nullptr, "pRotatE needs the modulus scalar");
    return MKB_OK;
}

inline bool mode_is_head(int mode) { return mode == MKB_MODE_HEAD; }

// Brackets a kernel launch with hipEvents on the launch stream when profiling of `kind` is enabled
// (mkb_profile_enable).  Usage:  { ProfScope ps(MKB_PROF_ADAM, st); hipLaunchKernelGGL(...); }
struct ProfScope {
    int kind;
    hipStream_t st;
    int slot;
    ProfScope(int kind, hipStream_t st);
    ~ProfScope();
};

// Layout of the gradient-seed matrix G the single-pass pooled backward reads: log2_blocks < 0 = plain [B, P]; otherwise
// tile-blocked in the kernel's slot order, G8[tile][block][half][lane][8 rows]: a lane of the backward fetches the 8 seeds of
// its slot with two 16-byte loads (the plain layout costs it 8 strided 4-byte loads per run).  blocks and halves are powers
// of two; position p = block + blocks * (lane * halves + half).
struct SeedLayout {
    int log2_blocks, log2_halves;
};
__host__ __device__ inline int64_t seed_index(const SeedLayout &L, int64_t i, int64_t p, int64_t P) {
    if (L.log2_blocks < 0) return i * P + p;
    const int64_t blocks = (int64_t)1 << L.log2_blocks, halves = (int64_t)1 << L.log2_halves;
    const int64_t pb = p & (blocks - 1), jj = p >> L.log2_blocks;
    const int64_t h = jj & (halves - 1), l = jj >> L.log2_halves;
    return (((((i >> 3) << L.log2_blocks) + pb) << L.log2_halves) + h) * 512 + l * 8 + (i & 7);
}

// The tail of a split-K GEMM that its caller folds into the NEXT launch instead of launching it (gemm_mfma.h):
//   kind 1: out[i] = c0 + c1 * sum_z part[z * n + i]                       (fixed order; consumed by the loss rows)
//   kind 2: out[c_idx[m]][col] += sum_z part[(z * M + m) * N + col]        (one atomic per element; rides the row backward)
//   kind 3: out[m][col] = c0 + c1 * sum_z part[(z * M + m) * N + col] for col < N only, rows of `out` are ldc long (the dense
//           prefix of the forward tile, score_pool_tile.h; consumed by the loss rows, which mask unused pairs to 0)
struct GemmTail {
    int kind;  // 0: nothing pending
    const float *part;
    float *out;
    const int64_t *c_idx;
    int M, N, nz;
    int64_t ldc, n;
    float c0, c1;
    // kind 2: rows m >= max(depth[0 .. n_depth)) were not computed (no batch row uses those pool positions): skipped
    const int *depth;
    int n_depth;
};

#ifdef __HIPCC__
// max of v[lo .. hi) over the workgroup (any size that is a multiple of 64, <= 1024 lanes); every lane gets the result.
// v[i] = one past the last pool position batch row i uses (written by the row kernels that open a pooled call): the GEMM
// route of the bilinear models multiplies only the pool positions somebody uses (P' ~ 340 of 512).
__device__ __forceinline__ int block_max_i32(const int *__restrict__ v, int lo, int hi, int *s_red) {
    int m = 0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += (int)blockDim.x) m = max(m, v[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    int r = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r = max(r, s_red[w]);
    __syncthreads();
    return r;
}
#endif

// loss.hip: Adversarial forward + gradient seeds.  defer_finish: the caller sums scratch[1 .. B] itself with
// adversarial_finish_block (mkb_pool_step: inside the row backward kernel, saving a launch).  neg_tail (kind 1): the
// scores are still split-K partials; the rows reduce them while they load them and write the final scores to neg_tail->out.
// occ (+ the batch's sample [B, 3] and pool [K columns]): occurrence counts of the batch's entities (RowStepArgs::occ).
int adversarial_launch(const float *pos, const float *neg, const float *weight, const uint16_t *cnt, int64_t B, int64_t K,
                       float alpha, const float *weight_sum, float *loss, float *dpos, float *dneg, float *scratch,
                       hipStream_t st, bool defer_finish, SeedLayout seeds = SeedLayout{-1, 0},
                       const GemmTail *neg_tail = nullptr, int *occ = nullptr,
                       const int64_t *occ_sample = nullptr, const int64_t *occ_pool = nullptr);

#ifdef __HIPCC__
// loss = -sum_i rowpart[i] / (2 W) by ONE 256-lane workgroup, fixed order (strided partial sums, wave64 butterfly, then
// the four wave totals left to right): bit-reproducible.  red: 4 floats of LDS.
__device__ __forceinline__ void adversarial_finish_block(const float *__restrict__ rowpart, int B, const float *__restrict__ scal,
                                                         float *__restrict__ loss, float *red) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) acc += rowpart[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = -0.5f * (red[0] + red[1] + red[2] + red[3]) / scal[0];
}
#endif

}  // namespace mkb

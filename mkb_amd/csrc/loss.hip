// mkb_adversarial: self-adversarial negative-sampling loss, forward AND gradient seed in one pass.
//
// Replaces losses.Adversarial.__call__ (losses/adversarial.py:21-30) and its autograd:
//   ps_i = logsigmoid(pos_i); p_ij = softmax_j(alpha * neg_ij) (detached); ns_i = sum_j p_ij logsigmoid(-neg_ij)
//   loss = ( -sum_i w_i ps_i / W  -  sum_i w_i ns_i / W ) / 2,   W = sum_i w_i
//   d loss/d pos_i = -(w_i / 2W) sigmoid(-pos_i);   d loss/d neg_ij = +(w_i / 2W) p_ij sigmoid(neg_ij)
// Optional column multiplicities cnt[i,j] (pooled path: column = pool position used cnt times by row i):
//   softmax and the sums run over columns weighted by cnt; dneg_ij is the summed gradient of its copies.
//
// One launch up to 8192 rows: every workgroup (one wave per row, 4 rows) first reduces W itself with the fixed tree
//   below (B / 256 loads per lane from L2; bit-identical in every workgroup), then row max / partition sum / weighted
//   log-sigmoid sum with wave64 shuffle reductions, writes dpos, dneg and the per-row partial.  A second single-
//   workgroup kernel sums the partials in a fixed order -- or, inside mkb_pool_step, the last workgroup of the row
//   backward kernel does (adversarial_finish_block), so the fused step spends ONE launch on the loss.
//   No float atomics: the loss is bit-reproducible run to run.
#include "common.h"
#include "model_math.h"

namespace mkb {

// (the first 256 lanes of the workgroup do the work: the same tree, hence the same bits, for 256- and 512-lane workgroups)
__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0 && wave < 4) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// the same sum with its first four strides' loads already made by the caller (v[z] = w[min(tid + 256 z, B - 1)], lanes < 256):
// the loss rows request them in front of their own loads
__device__ __forceinline__ float weight_sum_block_ahead(const float *__restrict__ w, int B, float *red, const float (&v)[4]) {
    float acc = 0.f;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int z = 0; z < 4; ++z) acc += (int)threadIdx.x + 256 * z < B ? v[z] : 0.f;
        for (int i0 = threadIdx.x + 1024; i0 < B; i0 += 1024) {
            float u[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) u[z] = w[min(i0 + 256 * z, B - 1)];
#pragma unroll
            for (int z = 0; z < 4; ++z) acc += i0 + 256 * z < B ? u[z] : 0.f;
        }
    }
    return block_sum_256(acc, red);
}

__device__ __forceinline__ float weight_sum_block(const float *__restrict__ w, int B, float *red) {
    // four strides' loads at a time (a loop with a run-time trip count is one round trip per iteration: every workgroup of the
    // loss launch walked 1024 weights in four of them); added in the order of the plain loop
    float acc = 0.f;
    if (threadIdx.x < 256)
        for (int i0 = threadIdx.x; i0 < B; i0 += 1024) {
            float v[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) v[z] = w[min(i0 + 256 * z, B - 1)];
#pragma unroll
            for (int z = 0; z < 4; ++z) acc += i0 + 256 * z < B ? v[z] : 0.f;
        }
    return block_sum_256(acc, red);
}

// scal[0] = W
__global__ __launch_bounds__(256) void weight_sum_kernel(const float *__restrict__ w, int B, float *__restrict__ scal) {
    __shared__ float red[4];
    const float acc = weight_sum_block(w, B, red);
    if (threadIdx.x == 0) scal[0] = acc;
}

// exp / log on the hardware transcendental unit (v_exp_f32 / v_log_f32, ~1 ulp of the base-2 function): the arguments
// here are scores of magnitude <= ~20, so exp's relative error stays ~1e-6 and log1p's absolute error ~1e-7 -- two
// orders below the 1e-4 / 1e-5 parity tolerances -- at a tenth of the instructions of the correctly rounded libm forms.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_log_sigmoid(float z) {  // min(z,0) - log1p(exp(-|z|)), the form ATen uses
    return fminf(z, 0.f) - 0.69314718055994531f * __builtin_amdgcn_logf(1.f + fast_exp(-fabsf(z)));
}
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + fast_exp(-z)); }

// one wave per row; rowpart[i] = w_i * (ps_i + ns_i).
// Columns are handled 64 at a time (column group t: lane l owns column l + 64 t).  Rows of up to kStageCols columns are read
// from global memory ONCE and staged in LDS -- scores in s_v, multiplicities (then softmax numerators) in s_e, both
// [column][rows + 1] words, so the 64 lanes of a group hit 64 different banks -- and the three passes (max, partition sums,
// gradient seeds) are LOOPS over the column groups.  (Round 3 kept the row in registers instead, which needs every pass fully
// unrolled over 16 groups x up to 16 split partials: 42 KB of straight-line code of which a wave executed ~9 KB once -- and
// per-launch code is paid per byte, the instruction cache is cold at every launch: DESIGN.md section 8.  The loads are issued
// four groups at a time, so a row still costs only two or three round trips.)  Longer rows re-read global memory.
// TILE (tile-blocked seed layout only): 512 lanes = the 8 rows of one row tile.  Written straight from the rows, the
// seeds of one row are 4-byte stores 2 KB apart (512 k partial-line writes per 1024 rows: the kernel's duration grew
// linearly with the rows, 12 -> 41 us from 1024 to 8192); staged through LDS (s_v itself) the tile goes out as whole 2 KB runs.
constexpr int kStageCols = 1024;
template <bool TILE>
__global__ __launch_bounds__(TILE ? 512 : 256) void adversarial_rows_kernel(const float *__restrict__ pos, const float *__restrict__ neg,
                                                               const float *__restrict__ w, const uint16_t *__restrict__ cnt,
                                                               int B, int K, float alpha, const float *__restrict__ scal,
                                                               float *__restrict__ scal_out, float *__restrict__ dpos,
                                                               float *__restrict__ dneg, float *__restrict__ rowpart, SeedLayout SL,
                                                               GemmTail NT,
                                                               int *__restrict__ occ, const int64_t *__restrict__ occ_sample,
                                                               const int64_t *__restrict__ occ_pool) {
    constexpr int NTH = TILE ? 512 : 256, RPB = NTH / 64, RS = RPB + 1;  // lanes and rows per workgroup, LDS words per column
    __shared__ float red[4];
    extern __shared__ float s_dyn[];
    const int cap = TILE ? (64 << (SL.log2_blocks + SL.log2_halves)) : 0;  // TILE: positions of the blocked layout (>= K)
    const bool staged = K <= kStageCols;
    float *s_v = s_dyn;                                  // TILE: [cap][9] (also the seed tile that goes out), else [K][5]
    float *s_e = s_dyn + (size_t)(TILE ? cap : K) * RS;  // [K][RS] (staged rows only)
    if constexpr (TILE) {
        // padding positions [K, cap) and rows past the batch must read as 0 in the tile that goes out; a full tile of a
        // pool that fills its layout (the headline shape) has neither: no clearing pass, no barrier (workgroup-uniform)
        if (K < cap || (int)(blockIdx.x + 1) * RPB > B) {
            for (int e = threadIdx.x; e < cap * RS; e += NTH) s_v[e] = 0.f;
            __syncthreads();
        }
    }
    // one lane per row counts the row's head and tail and its share of the pool ids: the ids are requested here and used behind
    // the row's own loads (requested, waited for and used up here they were one more round trip in front of everything else)
    const bool occ_lane = occ && (threadIdx.x & 63) == 0 && (int64_t)blockIdx.x * RPB + (threadIdx.x >> 6) < B;
    int64_t occ_h = 0, occ_t = 0, occ_p = -1;
    if (occ_lane) {
        const int64_t row = (int64_t)blockIdx.x * RPB + (threadIdx.x >> 6);
        occ_h = occ_sample[3 * row];
        occ_t = occ_sample[3 * row + 2];
        if (row < K) occ_p = occ_pool[row];
    }
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int i_raw = blockIdx.x * RPB + r;
    const int i = min(i_raw, B - 1);  // (rows past the batch load row B-1 and leave after the workgroup-wide W reduction)
    const bool live = i_raw < B;
    // everything else the row needs from global memory is requested here, in front of the row's own loads: the weights for W
    // (scal == nullptr), the row's weight and positive score -- each was a round trip of its own further down
    float wv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!scal && threadIdx.x < 256) {
#pragma unroll
        for (int z = 0; z < 4; ++z) wv[z] = w[min((int)threadIdx.x + 256 * z, B - 1)];
    }
    const float wi = w[i], p_i = pos[i];
    const float *nrow = neg + (int64_t)i * K;
    const uint16_t *crow = cnt ? cnt + (int64_t)i * K : nullptr;
    const int nt = (K + 63) >> 6;
    float m = -INFINITY;
    if (staged) {
        // scores still in split partials are reduced here, in the order of splitk_reduce_kernel (fixed: deterministic):
        //   kind 1: every column, part[z * n + i * K + j];  kind 3: columns < N (the forward tile's dense prefix), part[z][i][j]
        for (int t0 = 0; t0 < nt; t0 += 4) {
            // (the multiplicities stay raw until the second loop: converted here, each group's uint16 load was waited for in
            // front of the group's score loads -- the branch on `split` ends the basic block, and the wait with it: four round
            // trips per batch of "four groups together", eight per 512-column row)
            unsigned cu[4];
            float pz[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // the loads of four column groups are issued together
                const int t = t0 + u, j = lane + 64 * t, jc = min(j, K - 1);
                cu[u] = 1u;
                if (crow) cu[u] = crow[jc];
                const bool split = NT.kind == 1 || (NT.kind == 3 && 64 * t < NT.N);  // (wave-uniform: N is a multiple of 64)
                if (split) {
                    const float *pp = NT.kind == 1 ? NT.part + (int64_t)i * K + jc : NT.part + (int64_t)i * NT.N + min(j, NT.N - 1);
#pragma unroll
                    for (int z = 0; z < 8; ++z) pz[u][z] = pp[(int64_t)min(z, NT.nz - 1) * NT.n];
                } else {
                    pz[u][0] = nrow[jc];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u, j = lane + 64 * t;
                const bool ok = j < K;
                const bool split = NT.kind == 1 || (NT.kind == 3 && 64 * t < NT.N);
                const float c = ok ? (float)cu[u] : 0.f;
                float v = pz[u][0];
                if (split) {
                    float acc = 0.f;
#pragma unroll
                    for (int z = 0; z < 8; ++z) acc += z < NT.nz ? pz[u][z] : 0.f;
                    if (NT.nz > 8) {  // (up to 16 splits: the second eight, same order)
                        const float *pp = NT.kind == 1 ? NT.part + (int64_t)i * K + min(j, K - 1) : NT.part + (int64_t)i * NT.N + min(j, NT.N - 1);
                        for (int z = 8; z < NT.nz; ++z) acc += pp[(int64_t)z * NT.n];
                    }
                    v = NT.c0 + NT.c1 * acc;
                    if (NT.kind == 3) v = c > 0.f ? v : 0.f;  // (pairs the row does not use were computed: defined as 0)
                    if (ok) NT.out[(int64_t)i * K + j] = v;
                }
                v = ok ? v : 0.f;
                if (ok && live) {
                    s_v[j * RS + r] = v;
                    s_e[j * RS + r] = c;
                }
                if (c > 0.f) m = fmaxf(m, alpha * v);
            }
        }
    }
    if (occ_lane) {
        atomicAdd(occ + occ_h, 1);
        atomicAdd(occ + occ_t, 1);
        if (occ_p >= 0) atomicAdd(occ + occ_p, 1);
        for (int64_t p = (int64_t)blockIdx.x * RPB + (threadIdx.x >> 6) + B; p < K; p += B) atomicAdd(occ + occ_pool[p], 1);
    }
    // scal == nullptr: W is reduced here, by every workgroup alike (its loads and barriers run under the row's loads, which
    // were issued above); workgroup 0 publishes it for the finish step
    const float W = scal ? scal[0] : weight_sum_block_ahead(w, B, red, wv);
    if (!scal && blockIdx.x == 0 && threadIdx.x == 0) scal_out[0] = W;
    if (!TILE && !live) return;
    if (live) {
    if (!staged) {
        for (int j = lane; j < K; j += 64)
            if (!crow || crow[j]) m = fmaxf(m, alpha * nrow[j]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float z = 0.f, s = 0.f;
    if (staged) {
#pragma unroll 2
        for (int t = 0; t < nt; ++t) {
            const int j = lane + 64 * t;
            const float c = j < K ? s_e[j * RS + r] : 0.f;
            if (c > 0.f) {
                const float v = s_v[j * RS + r];
                const float e = c * fast_exp(alpha * v - m);
                z += e;
                s += e * fast_log_sigmoid(-v);
                s_e[j * RS + r] = e;  // keep the softmax numerator for the gradient pass
            }
        }
    } else {
        for (int j = lane; j < K; j += 64) {
            const float cc = crow ? (float)crow[j] : 1.f;
            if (cc > 0.f) {
                const float vv = nrow[j];
                const float e = cc * fast_exp(alpha * vv - m);
                z += e;
                s += e * fast_log_sigmoid(-vv);
            }
        }
    }
    z = wave_sum(z);
    s = wave_sum(s);
    const float coef = 0.5f * wi / W;
    const float invz = 1.f / z;
    if (staged) {
#pragma unroll 2
        for (int t = 0; t < nt; ++t) {
            const int j = lane + 64 * t;
            if (j < K) {
                const float e = s_e[j * RS + r];
                const float g = e > 0.f ? coef * (e * invz) * fast_sigmoid(s_v[j * RS + r]) : 0.f;
                if constexpr (TILE) s_v[j * RS + r] = g;
                else dneg[seed_index(SL, i, j, K)] = g;
            }
        }
    } else {
        for (int j = lane; j < K; j += 64) {
            const float cc = crow ? (float)crow[j] : 1.f;
            float g = 0.f;
            if (cc > 0.f) {
                const float vv = nrow[j];
                g = coef * (cc * fast_exp(alpha * vv - m) * invz) * fast_sigmoid(vv);
            }
            if constexpr (TILE) s_v[j * RS + r] = g;
            else dneg[seed_index(SL, i, j, K)] = g;
        }
    }
    if (lane == 0) {
        const float p = p_i;
        dpos[i] = -coef * fast_sigmoid(-p);
        rowpart[i] = wi * (fast_log_sigmoid(p) + s * invz);
    }
    }  // live
    if constexpr (TILE) {
        // the tile's region of the blocked layout is blocks * halves runs of 512 floats ([64 lanes][8 rows]); run c holds the
        // slots (l, c % halves) of position block c / halves: position p = pb + blocks * (l * halves + h)
        __syncthreads();
        const int blocks = 1 << SL.log2_blocks, halves = 1 << SL.log2_halves;
        float *tile_out = dneg + (int64_t)blockIdx.x * cap * 8;
        for (int q = threadIdx.x; q < cap * 2; q += NTH) {  // one float4 (4 rows of one slot) per step
            const int run = q >> 7, off = (q & 127) * 4, l = off >> 3, r0 = off & 7;
            const int pb = run >> SL.log2_halves, h = run & (halves - 1);
            const int p = pb + blocks * (l * halves + h);
            const float *src = s_v + p * RS + r0;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < cap) o = make_float4(src[0], src[1], src[2], src[3]);
            *reinterpret_cast<float4 *>(tile_out + (int64_t)q * 4) = o;
        }
    }
}

__global__ __launch_bounds__(256) void adversarial_finish_kernel(const float *__restrict__ rowpart, int B,
                                                                 const float *__restrict__ scal, float *__restrict__ loss) {
    __shared__ float red[4];
    adversarial_finish_block(rowpart, B, scal, loss, red);
}

int adversarial_launch(const float *pos, const float *neg, const float *weight, const uint16_t *cnt, int64_t B, int64_t K,
                       float alpha, const float *weight_sum, float *loss, float *dpos, float *dneg, float *scratch,
                       hipStream_t st, bool defer_finish, SeedLayout seeds, const GemmTail *neg_tail,
                       int *occ, const int64_t *occ_sample, const int64_t *occ_pool) {
    float *scal = scratch, *rowpart = scratch + 1;
    GemmTail nt{};
    if (neg_tail && (neg_tail->kind == 1 || neg_tail->kind == 3)) {
        if (K > kStageCols) return set_error(MKB_ERR_INVALID, "split-K scores can only ride rows of <= %d columns", kStageCols);
        nt = *neg_tail;
    }
    const float *scal_in = weight_sum;  // W of the whole (sharded) batch when the caller supplies it
    ProfScope ps(MKB_PROF_LOSS, st);
    if (!weight_sum && B > 8192) {      // too many rows for every workgroup to re-reduce W: one extra launch
        hipLaunchKernelGGL(weight_sum_kernel, dim3(1), dim3(256), 0, st, weight, (int)B, scal);
        scal_in = scal;
    }
    if (seeds.log2_blocks >= 0) {  // tile-blocked seeds: one row tile per 512-lane workgroup, staged through LDS
        const int cap = 64 << (seeds.log2_blocks + seeds.log2_halves);
        if (cap < K) return set_error(MKB_ERR_INVALID, "blocked seed layout holds %d positions, the rows have %d", cap, (int)K);
        // LDS: the seed tile [cap][9] (which first holds the scores) + the multiplicities [K][9] of rows short enough to stage
        const size_t lds = ((size_t)cap + (K <= kStageCols ? (size_t)K : 0)) * 9 * 4;
        static LdsOptIn lds_ok;  // more than 64 KB (pools of ~1000 positions and up): opt in once per device
        if (int rc = lds_ok.ensure(reinterpret_cast<const void *>(&adversarial_rows_kernel<true>), lds)) return rc;
        hipLaunchKernelGGL(adversarial_rows_kernel<true>, dim3((unsigned)((B + 7) / 8)), dim3(512), lds, st, pos, neg,
                           weight, cnt, (int)B, (int)K, alpha, scal_in, scal, dpos, dneg, rowpart, seeds, nt, occ,
                           occ_sample, occ_pool);
    } else {
        const size_t lds = K <= kStageCols ? (size_t)K * 5 * 4 * 2 : 0;  // scores + multiplicities [K][5] of staged rows
        hipLaunchKernelGGL(adversarial_rows_kernel<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), lds, st, pos, neg, weight, cnt,
                           (int)B, (int)K, alpha, scal_in, scal, dpos, dneg, rowpart, seeds, nt, occ, occ_sample,
                           occ_pool);
    }
    if (!defer_finish)
        hipLaunchKernelGGL(adversarial_finish_kernel, dim3(1), dim3(256), 0, st, rowpart, (int)B,
                           weight_sum ? weight_sum : scal, loss);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_adversarial(const float *pos, const float *neg, const float *weight, const uint16_t *cnt, int64_t B,
                               int64_t K, float alpha, const float *weight_sum, float *loss, float *dpos,
                               float *dneg, float *scratch, void *stream) {
    MKB_REQUIRE(pos && neg && weight && loss && dpos && dneg && scratch, "null pointer");
    MKB_REQUIRE(B > 0 && K > 0 && B <= INT32_MAX && K <= INT32_MAX, "bad B / K");
    return adversarial_launch(pos, neg, weight, cnt, B, K, alpha, weight_sum, loss, dpos, dneg, scratch, (hipStream_t)stream,
                              /*defer_finish=*/false);
}

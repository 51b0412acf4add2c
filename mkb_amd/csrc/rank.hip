// mkb_rank: filtered link-prediction rank of each test triple's target among ALL entities, on the device.
//
// Replaces evaluation.Evaluation.compute_score for head-/tail-batch (evaluation/evaluation.py:217-279) together
// with the candidate list / filter bias that datasets.base.TestDataset builds per item on the host
// (datasets/base.py:196-241): candidates are the n_entity ids in order; a candidate whose corrupted triple is
// ANOTHER true triple gets bias -100000 (i.e. can never outrank the target); rank = 1 + number of candidates
// scoring above the target (the reference reads it off a descending argsort, evaluation.py:245-262).
//
// Kernel 1  all_fwd : scores of B queries against every entity row.  Same lane-owns-dims tiling as the pooled
//           forward (8 rows per 1024-lane workgroup, swap/DPP wave reduction, LDS cross-wave combine per 16
//           candidates) but the candidates are simply consecutive table rows, split into slices over grid.y.
//           RotatE / TransE: the pooled forward's outer-product register tile instead; ComplEx / DistMult: one matrix-core
//           product (run_rank).
// Kernel 2  rank    : one wave per query: count scores above the target's, then walk the query's true set
//           (a contiguous range of the sorted key array) and take back the ones that were counted.
#include "common.h"
#include "model_math.h"
#include "score_pool_tile.h"  // the pooled forward's outer-product register tile, for the all-entity block of RotatE / TransE
#include "gemm_mfma.h"        // ... and the matrix-core product for ComplEx / DistMult, whose score is a dot product

#include <stdlib.h>
#include <algorithm>

namespace mkb {

constexpr int kWGr = 1024, kWavesR = 16, TIr = 8, kSlabR = 16;

struct AllArgs {
    const float *ent, *Q;
    float *S;  // [B, N]
    const float *modulus;
    int B, d;
    int64_t N, De;
    float kd, c0, c1;
};

__device__ __forceinline__ void reduce8_wave_r(const float (&v)[8], float &t0, float &t1) {
    float w[4], u[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 4]), false, false);
        w[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[j]), __float_as_uint(w[j + 2]), false, false);
        u[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float t = u[j];
        t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x128, 0xf, 0xf, false));
        t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x141, 0xf, 0xf, false));
        t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0xB1, 0xf, 0xf, false));
        t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x4E, 0xf, 0xf, false));
        u[j] = t;
    }
    t0 = u[0];
    t1 = u[1];
}

// NW waves per workgroup: 16 x KPT units per lane cover rows of up to 1024 KPT units; rows of <= 1024 units take 8 waves x 2 units
// per lane instead of 16 x 1 (round 5: the per-candidate wave reduction -- ~30 cross-lane operations -- is then paid once per 128
// units instead of once per 64, and the complex modulus runs on the packed pair form: the all-entity block of the headline
// model 4.5 -> 3.7 ms, the filtered evaluation of FB15k-237's test split 0.22 -> 0.19 s)
template <int MODEL, bool HEAD, int KPT, int NW = kWavesR>
__global__ __launch_bounds__(NW * 64) void all_fwd_kernel(AllArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    __shared__ float s_part[2][kSlabR][NW][TIr];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * TIr;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;
    float q0[TIr][KPT], q1[TIr][KPT];
#pragma unroll
    for (int r = 0; r < TIr; ++r)
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = (i0 + r < A.B) && (u0 + v < NU);
            const float *qrow = A.Q + (int64_t)min(i0 + r, A.B - 1) * A.De;
            const int uu = min(u0 + v, NU - 1);
            const float l0 = qrow[uu], l1 = CP ? qrow[A.d + uu] : 0.f;
            q0[r][v] = ok ? l0 : 0.f;
            q1[r][v] = ok ? l1 : 0.f;
        }
    const int64_t per = (A.N + gridDim.y - 1) / gridDim.y;
    const int64_t e_lo = (int64_t)blockIdx.y * per;
    const int64_t e_hi = min(A.N, e_lo + per);
    const int n_mine = (int)max((int64_t)0, e_hi - e_lo);
    for (int j = 0; j < n_mine; ++j) {
        const float *x = A.ent + (e_lo + j) * A.De;
        float x0[KPT], x1[KPT];
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = u0 + v < NU;
            const int uu = min(u0 + v, NU - 1);
            const float l0 = x[uu], l1 = CP ? x[A.d + uu] : 0.f;
            x0[v] = ok ? l0 : 0.f;
            x1[v] = ok ? l1 : 0.f;
        }
        float part[TIr];
#pragma unroll
        for (int r = 0; r < TIr; ++r) {
            part[r] = 0.f;
            if constexpr (CP && KPT % 2 == 0) {
                f2 acc = pair_term_cmod2(f2{q0[r][0], q0[r][1]}, f2{q1[r][0], q1[r][1]}, f2{x0[0], x0[1]}, f2{x1[0], x1[1]});
#pragma unroll
                for (int v = 2; v < KPT; v += 2)
                    acc += pair_term_cmod2(f2{q0[r][v], q0[r][v + 1]}, f2{q1[r][v], q1[r][v + 1]}, f2{x0[v], x0[v + 1]}, f2{x1[v], x1[v + 1]});
                part[r] = acc.x + acc.y;
            } else {
#pragma unroll
                for (int v = 0; v < KPT; ++v) {
                    if constexpr (CP) part[r] += pair_term_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]});
                    else part[r] += pair_term_real<MODEL, HEAD>(q0[r][v], x0[v], A.kd);
                }
            }
        }
        float t0, t1;
        reduce8_wave_r(part, t0, t1);
        const int jj = j % kSlabR, buf = (j / kSlabR) & 1;
        if ((lane & 15) == 0) {
            const int R = lane >> 4, r = 4 * (R >> 1) + 2 * (R & 1);
            s_part[buf][jj][wave][r] = t0;
            s_part[buf][jj][wave][r + 1] = t1;
        }
        if (jj == kSlabR - 1 || j == n_mine - 1) {
            __syncthreads();
            const int j0 = j - jj, nb = jj + 1;
            if (tid < nb * TIr) {
                const int cj = tid % nb, r = tid / nb;  // consecutive lanes -> consecutive candidates (coalesced store)
                if (i0 + r < A.B) {
                    float sum = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sum += s_part[buf][cj][w][r];
                    if constexpr (MODEL == MKB_PROTATE) sum *= A.modulus[0];
                    A.S[(int64_t)(i0 + r) * A.N + e_lo + j0 + cj] = A.c0 + A.c1 * sum;
                }
            }
        }
    }
}

// Position of the target in the reference's torch.argsort(score, descending=True) (evaluation.py:245-262).  Ties and NaN
// need care: "1 + #{S[e] > S[target]}" alone ranks the target FIRST whenever nothing compares greater, i.e. for a collapsed
// model (all scores equal) or a diverged one (NaN anywhere in the comparison) -- MRR = HITS@k = 1.0 for a broken run, which
// the early-stopping logic of Pipeline.learn would then keep.  The order used here is the one of a stable descending sort
// with torch's NaN convention: NaN sorts before every number, equal keys keep candidate order (lower entity id first).
__device__ __forceinline__ bool ranks_before(float a, int64_t ia, float b, int64_t ib) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return an && (!bn || ia < ib);
    return a > b || (a == b && ia < ib);
}

// one wave per query
// (c0, c1): S holds finished scores (0, 1) or the raw pair sums of the register-tile route, finished here as c0 + c1 * sum -- the
// same fp32 operation the all-entity kernel applies, so that ties fall as they do there
__global__ __launch_bounds__(256) void rank_kernel(const float *__restrict__ S, const int64_t *__restrict__ sample, int B,
                                                   int64_t N, int64_t R, int head_mode,
                                                   const int64_t *__restrict__ keys, int64_t nk,
                                                   int64_t *__restrict__ rank, float c0, float c1, int64_t ld) {  // ld: floats between rows of S
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const int64_t h = sample[3 * (int64_t)i], r = sample[3 * (int64_t)i + 1], t = sample[3 * (int64_t)i + 2];
    const int64_t target = head_mode ? h : t;
    const float *row = S + (int64_t)i * ld;
    auto val = [&](int64_t e) { return c0 + c1 * row[e]; };
    const float st = val(target);
    int64_t cnt = 0;
    // eight strides' loads at a time (a loop with a run-time trip count waits for every load before it issues the next: 227 round
    // trips per query at FB15k-237's size)
    for (int64_t e0 = lane; e0 < N; e0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = row[min(e0 + 64 * k, N - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t e = e0 + 64 * k;
            cnt += (e < N && ranks_before(c0 + c1 * v[k], e, st, target)) ? 1 : 0;
        }
    }
    // other true triples (base.py:213-216 / 229-232): take back those that were counted
    const int64_t base = ((head_mode ? t : h) * R + r) * N;  // keys of this (fixed entity, relation) pair are contiguous
    int64_t lo = 0, hi = nk;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < base) lo = mid + 1; else hi = mid; }
    for (int64_t k = lo + lane; k < nk; k += 64) {
        const int64_t key = keys[k];
        if (key >= base + N) break;
        const int64_t e = key - base;
        if (e != target && ranks_before(val(e), e, st, target)) cnt -= 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) rank[i] = cnt + 1;
}

// scores_out[i, e] = c0 + c1 * S[i * ld + e]: the finished score block the ranks were counted on, for callers that want the
// scores themselves (mkb_rank_scores)
__global__ __launch_bounds__(256) void export_scores_kernel(const float *__restrict__ S, float *__restrict__ out, int64_t N, int64_t ld,
                                                            float c0, float c1) {
    const int64_t i = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < N; e += (int64_t)gridDim.x * 256) out[i * N + e] = c0 + c1 * S[i * ld + e];
}

struct RowArgsR {
    const float *ent, *rel;
    const int64_t *sample;
    float *Q;
    int64_t De, Dr;
    int d;
    float kd;
};

template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void query_build_kernel_r(RowArgsR A) {
    const int64_t i = blockIdx.x;
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];
    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    float *q = A.Q + i * A.De;
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        const float *e = HEAD ? et : eh;
        for (int u = threadIdx.x; u < A.d; u += 256) {
            Cplx qq = build_q_cplx<MODEL, HEAD>(Cplx{e[u], e[A.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f}, A.kd);
            q[u] = qq.re;
            q[A.d + u] = qq.im;
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256)
            q[u] = build_q_real<MODEL, HEAD>(HEAD ? er[u] : eh[u], HEAD ? et[u] : er[u], A.kd);
    }
}

// ids[i] = min(i, n_real - 1) for i < n (n >= n_real: the padded tail repeats the last row)
__global__ __launch_bounds__(256) void iota_kernel(int64_t *ids, int64_t n, int64_t n_real) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids[i] = i < n_real ? i : n_real - 1;
}

template <int MODEL, bool HEAD>
static int run_rank(const mkb_tables_t *tb, const int64_t *sample, int64_t B, const int64_t *keys, int64_t nk, int64_t *rank,
                    float *Q, float *S, int64_t *ids, hipStream_t st, float *scores_out) {
    // the last launch(es) of every route: count the filtered ranks on S (and hand the finished scores out when asked)
    auto finish = [&](float f0, float f1, int64_t ld) {
        hipLaunchKernelGGL(rank_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, S, sample, (int)B, tb->n_entity,
                           tb->n_relation, HEAD ? 1 : 0, keys, nk, rank, f0, f1, ld);
        if (scores_out)
            hipLaunchKernelGGL(export_scores_kernel, dim3((unsigned)std::min<int64_t>((tb->n_entity + 255) / 256, 64), (unsigned)B), dim3(256), 0, st,
                               S, scores_out, tb->n_entity, ld, f0, f1);
    };
    RowArgsR ra{tb->ent, tb->rel, sample, Q, tb->entity_dim, tb->relation_dim, tb->hidden_dim, tb->phase_div};
    hipLaunchKernelGGL((query_build_kernel_r<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    const float c0 = ModelTraits<MODEL>::uses_gamma ? tb->gamma : 0.f, c1 = ModelTraits<MODEL>::uses_gamma ? -1.f : 1.f;
    // RotatE / TransE: the all-entity block on the pooled forward's register tile (score_pool_tile.h: 64 queries x 64 entities per
    // workgroup, 4 x 4 pairs per lane, operands staged through LDS -- no per-candidate wave reduction at all), the "pool" being
    // every entity in order and nothing masked; S then holds raw pair sums, finished by the rank kernel.  MKB_RANK_TILE=0: A/B.
    if constexpr (ModelTraits<MODEL>::cplx_pair || MODEL == MKB_TRANSE) {
        static const bool tile_off = getenv("MKB_RANK_TILE") && getenv("MKB_RANK_TILE")[0] == '0';
        const bool shape_ok = ModelTraits<MODEL>::cplx_pair ? (tb->hidden_dim % 4 == 0 && tb->hidden_dim >= 32)
                                                            : (tb->entity_dim % 4 == 0 && tb->entity_dim >= 64);
        if (!tile_off && shape_ok && (((uintptr_t)tb->ent | (uintptr_t)Q | (uintptr_t)S) & 15) == 0 && ids && tb->n_entity < (1 << 30)) {
            hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((tb->n_entity + 255) / 256)), dim3(256), 0, st, ids, tb->n_entity, tb->n_entity);
            PoolArgs P{};
            P.ent = tb->ent; P.Q = Q; P.pool = ids; P.B = (int)B; P.P = (int)tb->n_entity; P.d = tb->hidden_dim; P.De = tb->entity_dim;
            P.kd = tb->phase_div; P.c0 = c0; P.c1 = c1;
            TileArgs T{};
            T.part = S; T.Kd = (int)tb->n_entity; T.ks = 1;
            T.row_tiles = (int)((B + kTileRows - 1) / kTileRows); T.pos_tiles = (int)((tb->n_entity + kTilePos - 1) / kTilePos);
            hipLaunchKernelGGL((pool_fwd_tile_kernel<MODEL, HEAD, 2>), dim3((unsigned)(T.row_tiles * T.pos_tiles)), dim3(256), 0, st, P, T);
            finish(c0, c1, tb->n_entity);
            MKB_LAUNCH_CHECK();
            return MKB_OK;
        }
    }
    // ComplEx / DistMult: score = <q, x> over the entity row, so the all-entity block is ONE product Q [B, De] . E^T [De, N] on the
    // matrix cores (gemm_mfma.h: 128-row tiles, fp32 operands split three ways on the bf16 pipe as in the pooled forward of these
    // models).  N is padded to a multiple of 4 through the id list (the tail repeats the last row; the rank kernel never reads it),
    // S rows are Npad floats apart.  Ragged last batches (B % 4 != 0) keep the lane-owns-dims kernel.  MKB_RANK_GEMM=0: A/B.
    if constexpr (MODEL == MKB_COMPLEX || MODEL == MKB_DISTMULT) {
        static const bool gemm_off = getenv("MKB_RANK_GEMM") && getenv("MKB_RANK_GEMM")[0] == '0';
        const int64_t Npad = (tb->n_entity + 3) & ~(int64_t)3;
        if (!gemm_off && ids && B % 4 == 0 && tb->entity_dim % 4 == 0 && tb->entity_dim >= 16 && B >= 32 &&
            (((uintptr_t)tb->ent | (uintptr_t)Q | (uintptr_t)S) & 15) == 0 && Npad < (1 << 30)) {
            hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, st, ids, Npad, tb->n_entity);
            GemmArgs G{};
            G.A = Q; G.B = tb->ent; G.C = S; G.b_idx = ids; G.b_rows = tb->n_entity;
            G.M = (int)B; G.N = (int)Npad; G.K = (int)tb->entity_dim; G.ksplit = 1;
            G.lda = tb->entity_dim; G.ldb = tb->entity_dim; G.ldc = Npad; G.c0 = c0; G.c1 = c1;
            // max_ks = 1: the workspace holds ONE [B, Npad] score block -- a K split would write its partial products past it
            // (over the id list this very product gathers through).  `partials` = S only admits the 128-row tiles.
            if (int rc = launch_gemm<true, true, GEMM_STORE_AFFINE>(G, st, /*partials=*/S, nullptr, 0, 0, nullptr, /*max_ks=*/1)) return rc;
            finish(0.f, 1.f, Npad);
            MKB_LAUNCH_CHECK();
            return MKB_OK;
        }
    }
    AllArgs A{tb->ent, Q, S, tb->modulus, (int)B, tb->hidden_dim, tb->n_entity, tb->entity_dim, tb->phase_div, c0, c1};
    const int tiles = (int)((B + TIr - 1) / TIr);
    int slices = (512 + tiles - 1) / tiles;  // aim for >= 2 workgroups per CU
    if (slices < 1) slices = 1;
    if ((int64_t)slices > tb->n_entity) slices = (int)tb->n_entity;
    dim3 grid((unsigned)tiles, (unsigned)slices);
    const int NU = tb->model == MKB_ROTATE ? tb->hidden_dim : (int)tb->entity_dim;
    static const bool narrow_off = getenv("MKB_RANK_WIDE") != nullptr;  // A/B switch: 16 waves x 1 unit per lane as before
    if (NU <= kWGr / 2) hipLaunchKernelGGL((all_fwd_kernel<MODEL, HEAD, 1>), grid, dim3(kWGr), 0, st, A);
    else if (NU <= kWGr && !narrow_off) hipLaunchKernelGGL((all_fwd_kernel<MODEL, HEAD, 2, 8>), grid, dim3(512), 0, st, A);
    else if (NU <= kWGr) hipLaunchKernelGGL((all_fwd_kernel<MODEL, HEAD, 1>), grid, dim3(kWGr), 0, st, A);
    else if (NU <= 2 * kWGr) hipLaunchKernelGGL((all_fwd_kernel<MODEL, HEAD, 2>), grid, dim3(kWGr), 0, st, A);
    else hipLaunchKernelGGL((all_fwd_kernel<MODEL, HEAD, 4>), grid, dim3(kWGr), 0, st, A);
    finish(0.f, 1.f, tb->n_entity);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

}  // namespace mkb

using namespace mkb;

extern "C" int64_t mkb_rank_workspace_bytes(const mkb_tables_t *tb, int64_t B) {
    if (!tb || B <= 0) return 0;
    return (int64_t)(((size_t)B * tb->entity_dim * 4 + 255) & ~(size_t)255) + (int64_t)(((size_t)B * (tb->n_entity + 3) * 4 + 255) & ~(size_t)255) +
           (int64_t)(tb->n_entity + 3) * 8;
}

static int rank_impl(const mkb_tables_t *tb, const int64_t *sample, int64_t B, int mode, const int64_t *true_keys,
                     int64_t n_true, int64_t *rank, float *scores, void *ws, int64_t ws_bytes, void *stream) {
    if (int rc = validate_tables(tb)) return rc;
    MKB_REQUIRE(sample && rank && ws && (true_keys || n_true == 0), "null pointer");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "mkb_rank needs head-batch or tail-batch");
    MKB_REQUIRE(B > 0 && B <= INT32_MAX, "bad B");
    MKB_REQUIRE(ws_bytes >= mkb_rank_workspace_bytes(tb, B) && (((uintptr_t)ws) & 255) == 0, "workspace too small / unaligned");
    const int NU = tb->model == MKB_ROTATE ? tb->hidden_dim : (int)tb->entity_dim;
    MKB_REQUIRE(NU <= 4 * kWGr, "rows of more than 4096 units are not supported");
    float *Q = (float *)ws;
    float *S = (float *)((unsigned char *)ws + (((size_t)B * tb->entity_dim * 4 + 255) & ~(size_t)255));
    int64_t *ids = (int64_t *)((unsigned char *)S + (((size_t)B * (tb->n_entity + 3) * 4 + 255) & ~(size_t)255));  // [N + 3] 0, 1, ... (tile / GEMM routes)
    hipStream_t st = (hipStream_t)stream;
    const bool head = mode == MKB_MODE_HEAD;
    switch (tb->model) {
        case MKB_TRANSE: return head ? run_rank<MKB_TRANSE, true>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores) : run_rank<MKB_TRANSE, false>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores);
        case MKB_ROTATE: return head ? run_rank<MKB_ROTATE, true>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores) : run_rank<MKB_ROTATE, false>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores);
        case MKB_COMPLEX: return head ? run_rank<MKB_COMPLEX, true>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores) : run_rank<MKB_COMPLEX, false>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores);
        case MKB_DISTMULT: return head ? run_rank<MKB_DISTMULT, true>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores) : run_rank<MKB_DISTMULT, false>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores);
        case MKB_PROTATE: return head ? run_rank<MKB_PROTATE, true>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores) : run_rank<MKB_PROTATE, false>(tb, sample, B, true_keys, n_true, rank, Q, S, ids, st, scores);
    }
    return set_error(MKB_ERR_INVALID, "unknown model");
}

extern "C" int mkb_rank(const mkb_tables_t *tb, const int64_t *sample, int64_t B, int mode, const int64_t *true_keys,
                        int64_t n_true, int64_t *rank, void *ws, int64_t ws_bytes, void *stream) {
    return rank_impl(tb, sample, B, mode, true_keys, n_true, rank, nullptr, ws, ws_bytes, stream);
}

extern "C" int mkb_rank_scores(const mkb_tables_t *tb, const int64_t *sample, int64_t B, int mode, const int64_t *true_keys,
                               int64_t n_true, int64_t *rank, float *scores, void *ws, int64_t ws_bytes, void *stream) {
    MKB_REQUIRE(scores, "null pointer");
    return rank_impl(tb, sample, B, mode, true_keys, n_true, rank, scores, ws, ws_bytes, stream);
}

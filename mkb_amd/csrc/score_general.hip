// General (arbitrary candidate ids) scoring kernels: mkb_score_fwd / mkb_score_bwd.
//
// Replaces, for one call of model.forward(sample, negative_sample, mode) and its autograd:
//   BaseModel.batch gather (models/base.py:153-207) + <Model>.forward math (models/*.py) and the
//   index_select backward (dense index_add_), SURVEY.md a3-a8, a13.
//
// Forward  : one workgroup per batch row.  The row's query vector q_i (built from the two fixed
//            operands) is staged in LDS once and shared by all K candidates; each of the 4 waves streams
//            whole candidate rows with coalesced loads and finishes with a wave64 shuffle reduction.
// Backward : candidates given (K > 1 slots per row): two passes over the B*K pairs, lanes OWN units k in both (no cross-lane
//            reduction), neither writes a gradient per pair:
//              dq pass   one workgroup per batch row; dq accumulates in registers over the row's K candidates and is chained
//                        into the fixed operands' rows (one atomic per element and row);
//              dx pass   the pairs are SORTED BY CANDIDATE (hipCUB radix sort of B*K int32 keys); a workgroup walks 64
//                        consecutive sorted pairs, keeps the candidate row and its running gradient in registers and adds
//                        them to the table gradient only when the candidate changes: B*K/64 ... #distinct flushes instead
//                        of one fp32 atomic per (pair, dim) -- 524 M at the headline shape, which was ~2/3 of the step.
//            The pair term is evaluated in both passes (cheaper than the atomics it replaces).  Positive triples
//            (no candidates, K = 1) keep the one-pass kernel.
// The [B,K,D] intermediates of the reference are never materialised.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "model_math.h"

#include <stdlib.h>

namespace mkb {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

struct TablesDev {
    const float *ent, *rel, *modulus;
    int64_t De, Dr;
    int d;
    float gamma, kd;
    int vec4;  // rows (and their complex halves) are 16-byte aligned and a multiple of 4 units long
};

static TablesDev to_dev(const mkb_tables_t *tb) {
    TablesDev t;
    t.ent = tb->ent; t.rel = tb->rel; t.modulus = tb->modulus;
    t.De = tb->entity_dim; t.Dr = tb->relation_dim; t.d = tb->hidden_dim;
    t.gamma = tb->gamma; t.kd = tb->phase_div;
    const bool cp = tb->model == MKB_ROTATE;
    t.vec4 = ((((uintptr_t)tb->ent) & 15) == 0 && tb->entity_dim % 4 == 0 && (!cp || tb->hidden_dim % 4 == 0)) ? 1 : 0;
    return t;
}

// Build q for unit u of row (h, r, t).  Real models: one float.  Complex-query models: (re, im).
template <int MODEL, bool HEAD>
__device__ __forceinline__ void build_unit(const TablesDev &T, const float *eh, const float *er, const float *et,
                                           int u, float &q0, float &q1) {
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        const float *e = HEAD ? et : eh;
        Cplx ec{e[u], e[T.d + u]};
        Cplx rc{er[u], MODEL == MKB_COMPLEX ? er[T.d + u] : 0.f};
        Cplx q = build_q_cplx<MODEL, HEAD>(ec, rc, T.kd);
        q0 = q.re; q1 = q.im;
    } else {
        const float a = HEAD ? er[u] : eh[u];
        const float b = HEAD ? et[u] : er[u];
        q0 = build_q_real<MODEL, HEAD>(a, b, T.kd);
        q1 = 0.f;
    }
}

template <int MODEL, bool HEAD>
__global__ __launch_bounds__(kBlock) void score_fwd_kernel(TablesDev T, const int64_t *__restrict__ sample,
                                                           const int64_t *__restrict__ cand, int K,
                                                           float *__restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) float q_lds[];
    const int i = blockIdx.x;
    const int64_t h = sample[3 * (int64_t)i], r = sample[3 * (int64_t)i + 1], t = sample[3 * (int64_t)i + 2];
    const float *eh = T.ent + h * T.De, *er = T.rel + r * T.Dr, *et = T.ent + t * T.De;
    const int U = ModelTraits<MODEL>::cplx_query ? T.d : (int)T.De;
    for (int u = threadIdx.x; u < U; u += kBlock) {
        float q0, q1;
        build_unit<MODEL, HEAD>(T, eh, er, et, u, q0, q1);
        q_lds[u] = q0;
        if constexpr (ModelTraits<MODEL>::cplx_query) q_lds[T.d + u] = q1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float modulus = (MODEL == MKB_PROTATE) ? T.modulus[0] : 0.f;
    const bool vec4 = T.vec4 != 0;
    for (int j = wave; j < K; j += kWaves) {
        const int64_t c = cand ? cand[(int64_t)i * K + j] : t;
        const float *x = T.ent + c * T.De;
        float acc = 0.f;
        if constexpr (ModelTraits<MODEL>::cplx_pair) {
            if (vec4) {  // 16-byte loads of the candidate row halves and of the staged query
                for (int k = lane * 4; k < T.d; k += 256) {
                    const float4 xr = *reinterpret_cast<const float4 *>(x + k), xi = *reinterpret_cast<const float4 *>(x + T.d + k);
                    const float4 qr = *reinterpret_cast<const float4 *>(q_lds + k), qi = *reinterpret_cast<const float4 *>(q_lds + T.d + k);
                    const f2 t0 = pair_term_cmod2(f2{qr.x, qr.y}, f2{qi.x, qi.y}, f2{xr.x, xr.y}, f2{xi.x, xi.y});
                    const f2 t1 = pair_term_cmod2(f2{qr.z, qr.w}, f2{qi.z, qi.w}, f2{xr.z, xr.w}, f2{xi.z, xi.w});
                    acc += (t0.x + t0.y) + (t1.x + t1.y);
                }
            } else {
                for (int k = lane; k < T.d; k += 64)
                    acc += pair_term_cmod(Cplx{q_lds[k], q_lds[T.d + k]}, Cplx{x[k], x[T.d + k]});
            }
        } else {
            if (vec4) {
                for (int k = lane * 4; k < (int)T.De; k += 256) {
                    const float4 xv = *reinterpret_cast<const float4 *>(x + k), qv = *reinterpret_cast<const float4 *>(q_lds + k);
                    acc += (pair_term_real<MODEL, HEAD>(qv.x, xv.x, T.kd) + pair_term_real<MODEL, HEAD>(qv.y, xv.y, T.kd))
                           + (pair_term_real<MODEL, HEAD>(qv.z, xv.z, T.kd) + pair_term_real<MODEL, HEAD>(qv.w, xv.w, T.kd));
                }
            } else {
                for (int k = lane; k < (int)T.De; k += 64) acc += pair_term_real<MODEL, HEAD>(q_lds[k], x[k], T.kd);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) score[(int64_t)i * K + j] = finish_score<MODEL>(acc, T.gamma, modulus);
    }
}

template <int MODEL, bool HEAD>
__global__ __launch_bounds__(kBlock) void score_bwd_kernel(TablesDev T, mkb_grads_t G,
                                                           const int64_t *__restrict__ sample,
                                                           const int64_t *__restrict__ cand, int K,
                                                           const float *__restrict__ dscore) {
    __shared__ float red[kWaves];
    const int i = blockIdx.x;
    const int64_t h = sample[3 * (int64_t)i], r = sample[3 * (int64_t)i + 1], t = sample[3 * (int64_t)i + 2];
    const float *eh = T.ent + h * T.De, *er = T.rel + r * T.Dr, *et = T.ent + t * T.De;
    float *g_e = G.g_ent + (HEAD ? t : h) * T.De;  // fixed entity operand of the query
    float *g_r = G.g_rel + r * T.Dr;
    const int U = ModelTraits<MODEL>::cplx_query ? T.d : (int)T.De;
    const float modulus = (MODEL == MKB_PROTATE) ? T.modulus[0] : 0.f;
    float extra = 0.f;
    for (int u = threadIdx.x; u < U; u += kBlock) {
        float q0, q1;
        build_unit<MODEL, HEAD>(T, eh, er, et, u, q0, q1);
        float dq0 = 0.f, dq1 = 0.f;
        for (int j = 0; j < K; ++j) {
            const float g = dscore[(int64_t)i * K + j];
            const int64_t c = cand ? cand[(int64_t)i * K + j] : t;
            const float *x = T.ent + c * T.De;
            float *gx = G.g_ent + c * T.De;
            if constexpr (ModelTraits<MODEL>::cplx_pair) {
                Cplx dq, dx;
                pair_bwd_cmod(Cplx{q0, q1}, Cplx{x[u], x[T.d + u]}, g, dq, dx);
                dq0 += dq.re; dq1 += dq.im;
                atomicAdd(gx + u, dx.re);
                atomicAdd(gx + T.d + u, dx.im);
            } else if constexpr (ModelTraits<MODEL>::cplx_query) {  // ComplEx: dot over both halves
                float a, b, e0 = 0.f;
                pair_bwd_real<MODEL, HEAD>(q0, x[u], g, T.kd, modulus, a, b, e0);
                dq0 += a; atomicAdd(gx + u, b);
                pair_bwd_real<MODEL, HEAD>(q1, x[T.d + u], g, T.kd, modulus, a, b, e0);
                dq1 += a; atomicAdd(gx + T.d + u, b);
            } else {
                float a, b, e0 = 0.f;
                pair_bwd_real<MODEL, HEAD>(q0, x[u], g, T.kd, modulus, a, b, e0);
                dq0 += a; atomicAdd(gx + u, b);
                extra += g * e0;
            }
        }
        // chain dq into the fixed operands (duplicates across rows add: atomics)
        if constexpr (ModelTraits<MODEL>::cplx_query) {
            const float *e = HEAD ? et : eh;
            Cplx de, dr;
            query_bwd_cplx<MODEL, HEAD>(Cplx{e[u], e[T.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[T.d + u] : 0.f},
                                        Cplx{dq0, dq1}, T.kd, de, dr);
            atomicAdd(g_e + u, de.re);
            atomicAdd(g_e + T.d + u, de.im);
            atomicAdd(g_r + u, dr.re);
            if constexpr (MODEL == MKB_COMPLEX) atomicAdd(g_r + T.d + u, dr.im);
        } else {
            const float a = HEAD ? er[u] : eh[u], b = HEAD ? et[u] : er[u];
            float da, db;
            query_bwd_real<MODEL, HEAD>(a, b, dq0, T.kd, da, db);
            // tail-style: a = h, b = r ; head-style: a = r, b = t
            atomicAdd((HEAD ? g_r : g_e) + u, da);
            atomicAdd((HEAD ? g_e : g_r) + u, db);
        }
    }
    if constexpr (MODEL == MKB_PROTATE) {  // d score / d modulus = - sum_k |sin z|   (protate.py:91)
        extra = wave_sum(extra);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = extra;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < kWaves; ++w) s += red[w];
            atomicAdd(G.g_modulus, -s);
        }
    }
}

// ---- two-pass backward for explicit candidates ---------------------------------------------------------------------
// Q[i] = query of row i (units as build_unit gives them; complex-query models: re at u, im at d + u)
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(kBlock) void general_query_kernel(TablesDev T, const int64_t *__restrict__ sample, float *__restrict__ Q) {
    const int64_t i = blockIdx.x;
    const int64_t h = sample[3 * i], r = sample[3 * i + 1], t = sample[3 * i + 2];
    const float *eh = T.ent + h * T.De, *er = T.rel + r * T.Dr, *et = T.ent + t * T.De;
    const int U = ModelTraits<MODEL>::cplx_query ? T.d : (int)T.De;
    for (int u = threadIdx.x; u < U; u += kBlock) {
        float q0, q1;
        build_unit<MODEL, HEAD>(T, eh, er, et, u, q0, q1);
        Q[i * T.De + u] = q0;
        if constexpr (ModelTraits<MODEL>::cplx_query) Q[i * T.De + T.d + u] = q1;
    }
}

// dq pass: the one-pass kernel without its per-pair gradient writes
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(kBlock) void score_bwd_q_kernel(TablesDev T, mkb_grads_t G, const int64_t *__restrict__ sample,
                                                             const int64_t *__restrict__ cand, int K,
                                                             const float *__restrict__ dscore) {
    __shared__ float red[kWaves];
    const int i = blockIdx.x;
    const int64_t h = sample[3 * (int64_t)i], r = sample[3 * (int64_t)i + 1], t = sample[3 * (int64_t)i + 2];
    const float *eh = T.ent + h * T.De, *er = T.rel + r * T.Dr, *et = T.ent + t * T.De;
    float *g_e = G.g_ent + (HEAD ? t : h) * T.De;
    float *g_r = G.g_rel + r * T.Dr;
    const int U = ModelTraits<MODEL>::cplx_query ? T.d : (int)T.De;
    const float modulus = (MODEL == MKB_PROTATE) ? T.modulus[0] : 0.f;
    float extra = 0.f;
    for (int u = threadIdx.x; u < U; u += kBlock) {
        float q0, q1;
        build_unit<MODEL, HEAD>(T, eh, er, et, u, q0, q1);
        float dq0 = 0.f, dq1 = 0.f;
        for (int j = 0; j < K; ++j) {
            const float g = dscore[(int64_t)i * K + j];
            const float *x = T.ent + cand[(int64_t)i * K + j] * T.De;
            if constexpr (ModelTraits<MODEL>::cplx_pair) {
                Cplx dq, dx;
                pair_bwd_cmod(Cplx{q0, q1}, Cplx{x[u], x[T.d + u]}, g, dq, dx);
                dq0 += dq.re; dq1 += dq.im;
            } else if constexpr (ModelTraits<MODEL>::cplx_query) {
                float a, b, e0 = 0.f;
                pair_bwd_real<MODEL, HEAD>(q0, x[u], g, T.kd, modulus, a, b, e0);
                dq0 += a;
                pair_bwd_real<MODEL, HEAD>(q1, x[T.d + u], g, T.kd, modulus, a, b, e0);
                dq1 += a;
            } else {
                float a, b, e0 = 0.f;
                pair_bwd_real<MODEL, HEAD>(q0, x[u], g, T.kd, modulus, a, b, e0);
                dq0 += a;
                extra += g * e0;
            }
        }
        if constexpr (ModelTraits<MODEL>::cplx_query) {
            const float *e = HEAD ? et : eh;
            Cplx de, dr;
            query_bwd_cplx<MODEL, HEAD>(Cplx{e[u], e[T.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[T.d + u] : 0.f},
                                        Cplx{dq0, dq1}, T.kd, de, dr);
            atomicAdd(g_e + u, de.re);
            atomicAdd(g_e + T.d + u, de.im);
            atomicAdd(g_r + u, dr.re);
            if constexpr (MODEL == MKB_COMPLEX) atomicAdd(g_r + T.d + u, dr.im);
        } else {
            const float a = HEAD ? er[u] : eh[u], b = HEAD ? et[u] : er[u];
            float da, db;
            query_bwd_real<MODEL, HEAD>(a, b, dq0, T.kd, da, db);
            atomicAdd((HEAD ? g_r : g_e) + u, da);
            atomicAdd((HEAD ? g_e : g_r) + u, db);
        }
    }
    if constexpr (MODEL == MKB_PROTATE) {
        extra = wave_sum(extra);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = extra;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < kWaves; ++w) s += red[w];
            atomicAdd(G.g_modulus, -s);
        }
    }
}

// dx pass: workgroup = kChunkPairs consecutive pairs of the candidate-sorted pair list
constexpr int kChunkPairs = 64;
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(kBlock) void score_bwd_x_kernel(TablesDev T, mkb_grads_t G, const int *__restrict__ sorted_cand,
                                                             const int *__restrict__ sorted_pair, int n_pairs, int K,
                                                             const float *__restrict__ Q, const float *__restrict__ dscore) {
    const int lo = blockIdx.x * kChunkPairs, hi = min(n_pairs, lo + kChunkPairs);
    const int U = ModelTraits<MODEL>::cplx_query ? T.d : (int)T.De;
    const float modulus = (MODEL == MKB_PROTATE) ? T.modulus[0] : 0.f;
    for (int u = threadIdx.x; u < U; u += kBlock) {
        int cur = -1;
        float x0 = 0.f, x1 = 0.f, dx0 = 0.f, dx1 = 0.f;
        auto flush = [&]() {
            if (cur < 0) return;
            float *gx = G.g_ent + (int64_t)cur * T.De;
            atomicAdd(gx + u, dx0);
            if constexpr (ModelTraits<MODEL>::cplx_query) atomicAdd(gx + T.d + u, dx1);
        };
        for (int p = lo; p < hi; ++p) {
            const int c = sorted_cand[p];  // (uniform across the workgroup)
            if (c != cur) {
                flush();
                cur = c;
                const float *x = T.ent + (int64_t)c * T.De;
                x0 = x[u];
                x1 = ModelTraits<MODEL>::cplx_query ? x[T.d + u] : 0.f;
                dx0 = dx1 = 0.f;
            }
            const int pair = sorted_pair[p];
            const float g = dscore[pair];
            const float *q = Q + (int64_t)(pair / K) * T.De;
            if constexpr (ModelTraits<MODEL>::cplx_pair) {
                Cplx dq, dx;
                pair_bwd_cmod(Cplx{q[u], q[T.d + u]}, Cplx{x0, x1}, g, dq, dx);
                dx0 += dx.re; dx1 += dx.im;
            } else if constexpr (ModelTraits<MODEL>::cplx_query) {
                float a, b, e0 = 0.f;
                pair_bwd_real<MODEL, HEAD>(q[u], x0, g, T.kd, modulus, a, b, e0);
                dx0 += b;
                pair_bwd_real<MODEL, HEAD>(q[T.d + u], x1, g, T.kd, modulus, a, b, e0);
                dx1 += b;
            } else {
                float a, b, e0 = 0.f;
                pair_bwd_real<MODEL, HEAD>(q[u], x0, g, T.kd, modulus, a, b, e0);
                dx0 += b;
            }
        }
        flush();
    }
}

__global__ __launch_bounds__(256) void pair_keys_kernel(const int64_t *__restrict__ cand, int n, int *__restrict__ keys, int *__restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { keys[i] = (int)cand[i]; vals[i] = i; }
}

// Scratch of the two-pass backward: queries, the pair list twice (radix sort ping-pong) and the sort's own storage.
struct Bwd2Scratch {
    size_t q_bytes, l_bytes, sort_bytes, total;
    int bits;
};
static int bwd2_scratch(const mkb_tables_t *tb, int64_t B, int64_t K, Bwd2Scratch &S) {
    const int64_t n = B * K;
    S.bits = 1;
    while (S.bits < 31 && ((int64_t)1 << S.bits) < tb->n_entity) ++S.bits;
    S.sort_bytes = 0;
    MKB_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, S.sort_bytes, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                                                     (int *)nullptr, (int)n, 0, S.bits, (hipStream_t)0));
    S.q_bytes = ((size_t)B * tb->entity_dim * 4 + 255) & ~(size_t)255;
    S.l_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
    S.total = S.q_bytes + 4 * S.l_bytes + ((S.sort_bytes + 255) & ~(size_t)255);
    return MKB_OK;
}
static bool bwd2_applies(const mkb_tables_t *tb, const int64_t *cand, int64_t B, int64_t K) {
    static const bool one_pass = getenv("MKB_GENERAL_ONE_PASS") != nullptr;  // A/B: per-pair atomics
    return cand && K > 1 && B * K <= INT32_MAX && tb->n_entity <= INT32_MAX && !one_pass;
}

template <int MODEL>
static int launch_bwd2(const TablesDev &T, const mkb_tables_t *tb, const mkb_grads_t &G, const int64_t *sample, const int64_t *cand,
                       int64_t B, int K, bool head, const float *dscore, void *ws, hipStream_t st) {
    const int64_t n = B * K;
    Bwd2Scratch S;
    if (int rc = bwd2_scratch(tb, B, K, S)) return rc;
    unsigned char *buf = (unsigned char *)ws;  // caller-owned (mkb_score_bwd_workspace_bytes): nothing is allocated here
    const size_t q_bytes = S.q_bytes, l_bytes = S.l_bytes;
    size_t sort_bytes = S.sort_bytes;
    const int bits = S.bits;
    float *Q = (float *)buf;
    int *k_in = (int *)(buf + q_bytes), *v_in = (int *)(buf + q_bytes + l_bytes), *k_out = (int *)(buf + q_bytes + 2 * l_bytes),
        *v_out = (int *)(buf + q_bytes + 3 * l_bytes);
    void *tmp = buf + q_bytes + 4 * l_bytes;
    ProfScope ps(MKB_PROF_GENERAL_BWD, st);
    hipLaunchKernelGGL(pair_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cand, (int)n, k_in, v_in);
    MKB_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, sort_bytes, k_in, k_out, v_in, v_out, (int)n, 0, bits, st));
    const unsigned chunks = (unsigned)((n + kChunkPairs - 1) / kChunkPairs);
    if (head) {
        hipLaunchKernelGGL((general_query_kernel<MODEL, true>), dim3((unsigned)B), dim3(kBlock), 0, st, T, sample, Q);
        hipLaunchKernelGGL((score_bwd_q_kernel<MODEL, true>), dim3((unsigned)B), dim3(kBlock), 0, st, T, G, sample, cand, K, dscore);
        hipLaunchKernelGGL((score_bwd_x_kernel<MODEL, true>), dim3(chunks), dim3(kBlock), 0, st, T, G, k_out, v_out, (int)n, K, Q, dscore);
    } else {
        hipLaunchKernelGGL((general_query_kernel<MODEL, false>), dim3((unsigned)B), dim3(kBlock), 0, st, T, sample, Q);
        hipLaunchKernelGGL((score_bwd_q_kernel<MODEL, false>), dim3((unsigned)B), dim3(kBlock), 0, st, T, G, sample, cand, K, dscore);
        hipLaunchKernelGGL((score_bwd_x_kernel<MODEL, false>), dim3(chunks), dim3(kBlock), 0, st, T, G, k_out, v_out, (int)n, K, Q, dscore);
    }
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL>
static int launch_fwd(const TablesDev &T, const int64_t *sample, const int64_t *cand, int64_t B, int K, bool head,
                      float *score, hipStream_t st) {
    const size_t lds = (size_t)T.De * sizeof(float);
    ProfScope ps(MKB_PROF_GENERAL_FWD, st);
    if (head)
        hipLaunchKernelGGL((score_fwd_kernel<MODEL, true>), dim3((unsigned)B), dim3(kBlock), lds, st, T, sample, cand, K, score);
    else
        hipLaunchKernelGGL((score_fwd_kernel<MODEL, false>), dim3((unsigned)B), dim3(kBlock), lds, st, T, sample, cand, K, score);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL>
static int launch_bwd(const TablesDev &T, const mkb_grads_t &G, const int64_t *sample, const int64_t *cand, int64_t B,
                      int K, bool head, const float *dscore, hipStream_t st) {
    ProfScope ps(MKB_PROF_GENERAL_BWD, st);
    if (head)
        hipLaunchKernelGGL((score_bwd_kernel<MODEL, true>), dim3((unsigned)B), dim3(kBlock), 0, st, T, G, sample, cand, K, dscore);
    else
        hipLaunchKernelGGL((score_bwd_kernel<MODEL, false>), dim3((unsigned)B), dim3(kBlock), 0, st, T, G, sample, cand, K, dscore);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

static int check_call(const mkb_tables_t *tb, const int64_t *sample, const int64_t *cand, int64_t B, int64_t K, int mode) {
    if (int rc = validate_tables(tb)) return rc;
    MKB_REQUIRE(sample != nullptr && B >= 0 && K >= 1, "bad sample / B / K");
    MKB_REQUIRE(mode >= MKB_MODE_DEFAULT && mode <= MKB_MODE_TAIL, "bad mode %d", mode);
    MKB_REQUIRE(mode == MKB_MODE_DEFAULT ? (cand == nullptr && K == 1) : (cand != nullptr),
                "default mode takes no candidates (K=1); head/tail-batch need them");
    MKB_REQUIRE(tb->entity_dim * 4 <= 64 * 1024, "entity_dim too large for the LDS-staged query");
    MKB_REQUIRE(K <= INT32_MAX && B <= INT32_MAX, "B / K too large");
    return MKB_OK;
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_score_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *cand, int64_t B, int64_t K,
                             int mode, float *score, void *stream) {
    if (int rc = check_call(tb, sample, cand, B, K, mode)) return rc;
    MKB_REQUIRE(score != nullptr, "score is null");
    if (B == 0) return MKB_OK;
    const TablesDev T = to_dev(tb);
    hipStream_t st = (hipStream_t)stream;
    const bool head = mode_is_head(mode);
    switch (tb->model) {
        case MKB_TRANSE: return launch_fwd<MKB_TRANSE>(T, sample, cand, B, (int)K, head, score, st);
        case MKB_ROTATE: return launch_fwd<MKB_ROTATE>(T, sample, cand, B, (int)K, head, score, st);
        case MKB_COMPLEX: return launch_fwd<MKB_COMPLEX>(T, sample, cand, B, (int)K, head, score, st);
        case MKB_DISTMULT: return launch_fwd<MKB_DISTMULT>(T, sample, cand, B, (int)K, head, score, st);
        case MKB_PROTATE: return launch_fwd<MKB_PROTATE>(T, sample, cand, B, (int)K, head, score, st);
    }
    return set_error(MKB_ERR_INVALID, "unknown model");
}

extern "C" int64_t mkb_score_bwd_workspace_bytes(const mkb_tables_t *tb, int64_t B, int64_t K, int mode) {
    if (!tb || B <= 0 || K <= 1 || mode == MKB_MODE_DEFAULT) return 0;
    if (B * K > INT32_MAX || tb->n_entity > INT32_MAX) return 0;  // (the one-pass kernel: no scratch)
    Bwd2Scratch S;
    return bwd2_scratch(tb, B, K, S) ? -1 : (int64_t)S.total;
}

extern "C" int mkb_score_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *cand,
                             int64_t B, int64_t K, int mode, const float *dscore, void *ws, void *stream) {
    if (int rc = check_call(tb, sample, cand, B, K, mode)) return rc;
    MKB_REQUIRE(gr && gr->g_ent && gr->g_rel && dscore, "null gradient buffer");
    MKB_REQUIRE(tb->model != MKB_PROTATE || gr->g_modulus, "pRotatE needs g_modulus");
    if (B == 0) return MKB_OK;
    const TablesDev T = to_dev(tb);
    hipStream_t st = (hipStream_t)stream;
    const bool head = mode_is_head(mode);
    if (bwd2_applies(tb, cand, B, K)) {
        MKB_REQUIRE(ws != nullptr && (((uintptr_t)ws) & 255) == 0,
                    "mkb_score_bwd needs a 256-byte aligned workspace of mkb_score_bwd_workspace_bytes() bytes");
        switch (tb->model) {
            case MKB_TRANSE: return launch_bwd2<MKB_TRANSE>(T, tb, *gr, sample, cand, B, (int)K, head, dscore, ws, st);
            case MKB_ROTATE: return launch_bwd2<MKB_ROTATE>(T, tb, *gr, sample, cand, B, (int)K, head, dscore, ws, st);
            case MKB_COMPLEX: return launch_bwd2<MKB_COMPLEX>(T, tb, *gr, sample, cand, B, (int)K, head, dscore, ws, st);
            case MKB_DISTMULT: return launch_bwd2<MKB_DISTMULT>(T, tb, *gr, sample, cand, B, (int)K, head, dscore, ws, st);
            case MKB_PROTATE: return launch_bwd2<MKB_PROTATE>(T, tb, *gr, sample, cand, B, (int)K, head, dscore, ws, st);
        }
    }
    switch (tb->model) {
        case MKB_TRANSE: return launch_bwd<MKB_TRANSE>(T, *gr, sample, cand, B, (int)K, head, dscore, st);
        case MKB_ROTATE: return launch_bwd<MKB_ROTATE>(T, *gr, sample, cand, B, (int)K, head, dscore, st);
        case MKB_COMPLEX: return launch_bwd<MKB_COMPLEX>(T, *gr, sample, cand, B, (int)K, head, dscore, st);
        case MKB_DISTMULT: return launch_bwd<MKB_DISTMULT>(T, *gr, sample, cand, B, (int)K, head, dscore, st);
        case MKB_PROTATE: return launch_bwd<MKB_PROTATE>(T, *gr, sample, cand, B, (int)K, head, dscore, st);
    }
    return set_error(MKB_ERR_INVALID, "unknown model");
}

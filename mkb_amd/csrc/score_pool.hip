// Pooled scoring kernels: the fast path of the training step (mkb_pool_step / mkb_pool_score_fwd).
//
// mkb's sampler draws ONE pool of P = 2K candidate entities per batch and every row filters that same pool
// (sampling/negative_sampling.py:166 is outside the per-row loop at :168).  So the B x K negative block of
// compose/pipeline.py:230-232 is really "B queries x (<= P) shared candidate rows": instead of gathering
// B*K entity rows (2.1 GB at the headline config, models/base.py:193-207) each candidate row is loaded once
// per TILE of rows and reused from registers, and its gradient is reduced over the tile in registers before
// it leaves the CU.
//
// Mapping (both kernels): a workgroup owns a tile of TI batch rows and a chunk of the embedding dimension;
// each LANE OWNS KPT units k of that chunk (unit = one complex number for RotatE, one float otherwise) and
// keeps q[TI][KPT] (and dq[TI][KPT] in backward) in registers.  It then walks the pool positions that at
// least one row of the tile uses (compacted list in LDS); whether row r of the tile uses position p is a
// wave-uniform bit, so unused (row, position) pairs are skipped by scalar branches, not masked lanes.
//   forward : per position, TI per-lane partial sums -> transposed wave64 shuffle reduction -> fp32 atomics
//             into the [B, P] sum buffer (the only cross-workgroup reduction: over the dimension chunks).
//   backward: no cross-lane traffic at all.  dq accumulates in registers over all positions (stored once),
//             the candidate-row gradient is summed over the tile's rows in registers and leaves as one
//             coalesced fp32 atomic per lane into the [P, De] pool-gradient buffer.
// VALU-bound by design (RotatE: sqrt/rsqrt per (row, slot, complex dim)); HBM sees each touched row once.
#include "common.h"
#include "model_math.h"

namespace mkb {

struct PoolArgs {
    const float *ent;      // [N, De]
    const float *Q;        // [B, De] queries
    const int64_t *pool;   // [P]
    const uint16_t *cnt;   // [B, P] multiplicity (0 = row does not use the position)
    const float *G;        // [B, P] d loss / d score (backward)
    float *S;              // [B, P] running sums (forward, atomics)
    float *dQ;             // [B, De] (backward)
    float *GX;             // [P, De] pool-row gradients (backward, atomics)
    float *g_modulus;      // pRotatE
    const float *modulus;  // pRotatE
    int B, P, d;
    int64_t De;
    float kd;
};

constexpr int kPoolBlock = 256;

// Compact the pool positions used by at least one of the tile's rows into LDS:
//   plist[a] = position, pmask[a] = bit r set iff row r uses it, prow[a] = entity id.  Returns the count.
template <int TI>
__device__ __forceinline__ int build_tile_list(const PoolArgs &A, int i0, int *plist, unsigned *pmask, int *prow,
                                               int *s_count) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int wave_cnt[kPoolBlock / 64];
    int n = 0;
    for (int base = 0; base < A.P; base += kPoolBlock) {
        const int p = base + tid;
        unsigned m = 0;
        if (p < A.P) {
#pragma unroll
            for (int r = 0; r < TI; ++r)
                if (i0 + r < A.B && A.cnt[(int64_t)(i0 + r) * A.P + p] != 0) m |= 1u << r;
        }
        const unsigned long long b = __ballot(m != 0);
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int off = n;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        int tot = 0;
        for (int w = 0; w < kPoolBlock / 64; ++w) tot += wave_cnt[w];
        if (m != 0) {
            const int a = off + __popcll(b & ((1ull << lane) - 1ull));
            plist[a] = p;
            pmask[a] = m;
            prow[a] = (int)A.pool[p];
        }
        n += tot;
        __syncthreads();
    }
    if (tid == 0) *s_count = n;
    __syncthreads();
    return *s_count;
}

// TI per-lane partial sums -> per-row wave totals.  After the call, lane L with (L & 7) == 0 holds in `out`
// the total of row  r = 4*bit5(L) + 2*bit4(L) + bit3(L)  (TI == 8), i.e. 8 result lanes per wave.
__device__ __forceinline__ float reduce8_transposed(const float (&v)[8], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    float w[4], u[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float send = b5 ? v[j] : v[j + 4];
        const float keep = b5 ? v[j + 4] : v[j];
        w[j] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float send = b4 ? w[j] : w[j + 2];
        const float keep = b4 ? w[j + 2] : w[j];
        u[j] = keep + __shfl_xor(send, 16, 64);
    }
    const float send = b3 ? u[0] : u[1];
    const float keep = b3 ? u[1] : u[0];
    float t = keep + __shfl_xor(send, 8, 64);
    t += __shfl_xor(t, 4, 64);
    t += __shfl_xor(t, 2, 64);
    t += __shfl_xor(t, 1, 64);
    return t;
}

// ------------------------------------------------------------------------------------------------ forward
template <int MODEL, bool HEAD, int TI, int KPT>
__global__ __launch_bounds__(kPoolBlock) void pool_fwd_kernel(PoolArgs A) {
    static_assert(TI == 8, "the transposed reduction is written for 8 rows per tile");
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int *prow = reinterpret_cast<int *>(lds_raw);  // all LDS words are 4-byte: immune to the static-LDS base shift
    int *plist = prow + A.P;
    unsigned *pmask = reinterpret_cast<unsigned *>(plist + A.P);
    __shared__ int s_count;

    const int i0 = blockIdx.x * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = (blockIdx.y * kPoolBlock + threadIdx.x) * KPT;
    const int lane = threadIdx.x & 63;
    const int n_act = build_tile_list<TI>(A, i0, plist, pmask, prow, &s_count);

    float q0[TI][KPT], q1[TI][KPT];
#pragma unroll
    for (int r = 0; r < TI; ++r)
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = (i0 + r < A.B) && (u0 + v < NU);
            const float *qrow = A.Q + (int64_t)(i0 + r) * A.De;
            q0[r][v] = ok ? qrow[u0 + v] : 0.f;
            q1[r][v] = (CP && ok) ? qrow[A.d + u0 + v] : 0.f;
        }

    for (int a = 0; a < n_act; ++a) {
        const unsigned m = __builtin_amdgcn_readfirstlane(pmask[a]);
        const int p = __builtin_amdgcn_readfirstlane(plist[a]);
        const float *x = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(prow[a]) * A.De;
        float x0[KPT], x1[KPT];
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = u0 + v < NU;
            x0[v] = ok ? x[u0 + v] : 0.f;
            x1[v] = (CP && ok) ? x[A.d + u0 + v] : 0.f;
        }
        float part[TI];
#pragma unroll
        for (int r = 0; r < TI; ++r) {
            part[r] = 0.f;
            if (m & (1u << r)) {
#pragma unroll
                for (int v = 0; v < KPT; ++v) {  // out-of-range units hold q = x = 0 and contribute exactly 0
                    if constexpr (CP) part[r] += pair_term_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]});
                    else part[r] += pair_term_real<MODEL, HEAD>(q0[r][v], x0[v], A.kd);
                }
            }
        }
        const float tot = reduce8_transposed(part, lane);
        if ((lane & 7) == 0) {
            const int r = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
            if (m & (1u << r)) atomicAdd(A.S + (int64_t)(i0 + r) * A.P + p, tot);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <int MODEL, bool HEAD, int TI, int KPT>
__global__ __launch_bounds__(kPoolBlock) void pool_bwd_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int *prow = reinterpret_cast<int *>(lds_raw);
    int *plist = prow + A.P;
    unsigned *pmask = reinterpret_cast<unsigned *>(plist + A.P);
    float *Gt = reinterpret_cast<float *>(pmask + A.P);  // [n_act][TI] gradient seeds of the tile
    __shared__ int s_count;
    __shared__ float red[kPoolBlock / 64];

    const int i0 = blockIdx.x * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = (blockIdx.y * kPoolBlock + threadIdx.x) * KPT;
    const int n_act = build_tile_list<TI>(A, i0, plist, pmask, prow, &s_count);
    for (int e = threadIdx.x; e < n_act * TI; e += kPoolBlock) {
        const int a = e / TI, r = e % TI;
        Gt[e] = (i0 + r < A.B) ? A.G[(int64_t)(i0 + r) * A.P + plist[a]] : 0.f;
    }
    __syncthreads();

    float q0[TI][KPT], q1[TI][KPT], dq0[TI][KPT], dq1[TI][KPT];
#pragma unroll
    for (int r = 0; r < TI; ++r)
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = (i0 + r < A.B) && (u0 + v < NU);
            const float *qrow = A.Q + (int64_t)(i0 + r) * A.De;
            q0[r][v] = ok ? qrow[u0 + v] : 0.f;
            q1[r][v] = (CP && ok) ? qrow[A.d + u0 + v] : 0.f;
            dq0[r][v] = 0.f;
            dq1[r][v] = 0.f;
        }
    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;
    float extra = 0.f;

    for (int a = 0; a < n_act; ++a) {
        const unsigned m = __builtin_amdgcn_readfirstlane(pmask[a]);
        const int p = __builtin_amdgcn_readfirstlane(plist[a]);
        const float *x = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(prow[a]) * A.De;
        float x0[KPT], x1[KPT], dx0[KPT], dx1[KPT];
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = u0 + v < NU;
            x0[v] = ok ? x[u0 + v] : 0.f;
            x1[v] = (CP && ok) ? x[A.d + u0 + v] : 0.f;
            dx0[v] = 0.f;
            dx1[v] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < TI; ++r) {
            if (m & (1u << r)) {
                const float g = Gt[a * TI + r];
#pragma unroll
                for (int v = 0; v < KPT; ++v) {
                    if constexpr (CP) {
                        Cplx dq, dx;
                        pair_bwd_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]}, g, dq, dx);
                        dq0[r][v] += dq.re; dq1[r][v] += dq.im;
                        dx0[v] += dx.re; dx1[v] += dx.im;
                    } else {
                        float dq, dx, e0 = 0.f;
                        pair_bwd_real<MODEL, HEAD>(q0[r][v], x0[v], g, A.kd, modulus, dq, dx, e0);
                        dq0[r][v] += dq;
                        dx0[v] += dx;
                        extra += g * e0;
                    }
                }
            }
        }
        float *gx = A.GX + (int64_t)p * A.De;
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if (u0 + v < NU) {
                atomicAdd(gx + u0 + v, dx0[v]);
                if constexpr (CP) atomicAdd(gx + A.d + u0 + v, dx1[v]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TI; ++r)
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if ((i0 + r < A.B) && (u0 + v < NU)) {
                float *dqrow = A.dQ + (int64_t)(i0 + r) * A.De;
                dqrow[u0 + v] = dq0[r][v];
                if constexpr (CP) dqrow[A.d + u0 + v] = dq1[r][v];
            }
        }
    if constexpr (MODEL == MKB_PROTATE) {
        extra = wave_sum(extra);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = extra;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < kPoolBlock / 64; ++w) s += red[w];
            atomicAdd(A.g_modulus, -s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ row kernels
struct RowArgs {
    const float *ent, *rel;
    const int64_t *sample;
    float *Q;          // [B, De] out (build) / dQ in (backward)
    float *g_ent, *g_rel;
    int64_t De, Dr;
    int d;
    float kd;
};

// Q[i] = query of row i (same code as the LDS staging of the general forward kernel, written to global)
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void query_build_kernel(RowArgs A) {
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

// chain dQ[i] into the fixed operands' gradient rows (duplicate rows add: atomics)
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void query_bwd_kernel(RowArgs A) {
    const int64_t i = blockIdx.x;
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];
    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    const float *dq = A.Q + i * A.De;
    float *g_e = A.g_ent + (HEAD ? t : h) * A.De;
    float *g_r = A.g_rel + r * A.Dr;
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        const float *e = HEAD ? et : eh;
        for (int u = threadIdx.x; u < A.d; u += 256) {
            Cplx de, dr;
            query_bwd_cplx<MODEL, HEAD>(Cplx{e[u], e[A.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f},
                                        Cplx{dq[u], dq[A.d + u]}, A.kd, de, dr);
            atomicAdd(g_e + u, de.re);
            atomicAdd(g_e + A.d + u, de.im);
            atomicAdd(g_r + u, dr.re);
            if constexpr (MODEL == MKB_COMPLEX) atomicAdd(g_r + A.d + u, dr.im);
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256) {
            float da, db;
            query_bwd_real<MODEL, HEAD>(HEAD ? er[u] : eh[u], HEAD ? et[u] : er[u], dq[u], A.kd, da, db);
            atomicAdd((HEAD ? g_r : g_e) + u, da);
            atomicAdd((HEAD ? g_e : g_r) + u, db);
        }
    }
}

// g_ent[pool[p]] += GX[p]  (pool positions may repeat an entity: atomics)
__global__ __launch_bounds__(256) void pool_scatter_kernel(const float *__restrict__ GX, const int64_t *__restrict__ pool,
                                                           float *__restrict__ g_ent, int64_t De) {
    const int64_t p = blockIdx.x;
    const float *src = GX + p * De;
    float *dst = g_ent + pool[p] * De;
    for (int64_t k = threadIdx.x; k < De; k += 256) {
        const float v = src[k];
        if (v != 0.f) atomicAdd(dst + k, v);
    }
}

// S <- final score where cnt > 0:  c0 + c1 * sum  (gamma - sum | sum | gamma - modulus * sum)
__global__ __launch_bounds__(256) void finish_scores_kernel(float *__restrict__ S, int64_t n, float c0, float c1,
                                                            const float *__restrict__ modulus) {
    const float scale = modulus ? c1 * modulus[0] : c1;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) S[e] = c0 + scale * S[e];
}

// ------------------------------------------------------------------------------------------------ host side
struct Workspace {
    float *Q, *dQ, *G, *GX, *dpos, *scratch;
    size_t bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static Workspace carve(void *ws, int64_t B, int64_t P, int64_t De) {
    Workspace w;
    unsigned char *p = (unsigned char *)ws;
    size_t off = 0;
    auto take = [&](size_t n) { void *r = p ? p + off : nullptr; off += align256(n); return (float *)r; };
    w.Q = take((size_t)B * De * 4);
    w.dQ = take((size_t)B * De * 4);
    w.G = take((size_t)B * P * 4);
    w.GX = take((size_t)P * De * 4);
    w.dpos = take((size_t)B * 4);
    w.scratch = take((size_t)(B + 1) * 4);
    w.bytes = off;
    return w;
}

template <int MODEL, bool HEAD>
static int run_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                   int64_t P, float *S, const Workspace &w, hipStream_t st) {
    constexpr int TI = 8, KPT = 1;
    RowArgs ra{tb->ent, tb->rel, sample, w.Q, nullptr, nullptr, tb->entity_dim, tb->relation_dim, tb->hidden_dim, tb->phase_div};
    hipLaunchKernelGGL((query_build_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    MKB_CHECK_HIP(hipMemsetAsync(S, 0, (size_t)B * P * 4, st));
    PoolArgs A{};
    A.ent = tb->ent; A.Q = w.Q; A.pool = pool; A.cnt = cnt; A.S = S; A.B = (int)B; A.P = (int)P; A.d = tb->hidden_dim;
    A.De = tb->entity_dim; A.kd = tb->phase_div; A.modulus = tb->modulus;
    const int NU = ModelTraits<MODEL>::cplx_pair ? tb->hidden_dim : (int)tb->entity_dim;
    dim3 grid((unsigned)((B + TI - 1) / TI), (unsigned)((NU + kPoolBlock * KPT - 1) / (kPoolBlock * KPT)));
    const size_t lds = (size_t)P * (4 + 4 + 4);
    hipLaunchKernelGGL((pool_fwd_kernel<MODEL, HEAD, TI, KPT>), grid, dim3(kPoolBlock), lds, st, A);
    float c0 = 0.f, c1 = 1.f;
    if (ModelTraits<MODEL>::uses_gamma) { c0 = tb->gamma; c1 = -1.f; }
    hipLaunchKernelGGL(finish_scores_kernel, dim3(512), dim3(256), 0, st, S, B * P, c0, c1,
                       MODEL == MKB_PROTATE ? tb->modulus : nullptr);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int run_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                   const uint16_t *cnt, int64_t B, int64_t P, const Workspace &w, hipStream_t st) {
    constexpr int TI = 8, KPT = 1;
    MKB_CHECK_HIP(hipMemsetAsync(w.GX, 0, (size_t)P * tb->entity_dim * 4, st));
    PoolArgs A{};
    A.ent = tb->ent; A.Q = w.Q; A.pool = pool; A.cnt = cnt; A.G = w.G; A.dQ = w.dQ; A.GX = w.GX; A.B = (int)B; A.P = (int)P;
    A.d = tb->hidden_dim; A.De = tb->entity_dim; A.kd = tb->phase_div; A.modulus = tb->modulus; A.g_modulus = gr->g_modulus;
    const int NU = ModelTraits<MODEL>::cplx_pair ? tb->hidden_dim : (int)tb->entity_dim;
    dim3 grid((unsigned)((B + TI - 1) / TI), (unsigned)((NU + kPoolBlock * KPT - 1) / (kPoolBlock * KPT)));
    const size_t lds = (size_t)P * (4 + 4 + 4 + 4 * TI);
    hipLaunchKernelGGL((pool_bwd_kernel<MODEL, HEAD, TI, KPT>), grid, dim3(kPoolBlock), lds, st, A);
    RowArgs ra{tb->ent, tb->rel, sample, w.dQ, gr->g_ent, gr->g_rel, tb->entity_dim, tb->relation_dim, tb->hidden_dim, tb->phase_div};
    hipLaunchKernelGGL((query_bwd_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    hipLaunchKernelGGL(pool_scatter_kernel, dim3((unsigned)P), dim3(256), 0, st, w.GX, pool, gr->g_ent, tb->entity_dim);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

#define MKB_DISPATCH(fn, model, head, ...)                                                            \
    switch (model) {                                                                                  \
        case MKB_TRANSE: return (head) ? fn<MKB_TRANSE, true>(__VA_ARGS__) : fn<MKB_TRANSE, false>(__VA_ARGS__);       \
        case MKB_ROTATE: return (head) ? fn<MKB_ROTATE, true>(__VA_ARGS__) : fn<MKB_ROTATE, false>(__VA_ARGS__);       \
        case MKB_COMPLEX: return (head) ? fn<MKB_COMPLEX, true>(__VA_ARGS__) : fn<MKB_COMPLEX, false>(__VA_ARGS__);    \
        case MKB_DISTMULT: return (head) ? fn<MKB_DISTMULT, true>(__VA_ARGS__) : fn<MKB_DISTMULT, false>(__VA_ARGS__); \
        case MKB_PROTATE: return (head) ? fn<MKB_PROTATE, true>(__VA_ARGS__) : fn<MKB_PROTATE, false>(__VA_ARGS__);    \
    }                                                                                                 \
    return set_error(MKB_ERR_INVALID, "unknown model")

static int dispatch_fwd(const mkb_tables_t *tb, bool head, const int64_t *sample, const int64_t *pool, const uint16_t *cnt,
                        int64_t B, int64_t P, float *S, const Workspace &w, hipStream_t st) {
    MKB_DISPATCH(run_fwd, tb->model, head, tb, sample, pool, cnt, B, P, S, w, st);
}
static int dispatch_bwd(const mkb_tables_t *tb, bool head, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                        const uint16_t *cnt, int64_t B, int64_t P, const Workspace &w, hipStream_t st) {
    MKB_DISPATCH(run_bwd, tb->model, head, tb, gr, sample, pool, cnt, B, P, w, st);
}

template <int MODEL, bool HEAD>
static int run_query_build(const RowArgs &ra, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL((query_build_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}
static int dispatch_query_build(const mkb_tables_t *tb, bool head, const RowArgs &ra, int64_t B, hipStream_t st) {
    MKB_DISPATCH(run_query_build, tb->model, head, ra, B, st);
}

static int check_pool_call(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                           int64_t K, int mode, const void *ws) {
    if (int rc = validate_tables(tb)) return rc;
    MKB_REQUIRE(sample && pool && cnt && ws, "null pointer");
    MKB_REQUIRE(B > 0 && K > 0 && B <= INT32_MAX, "bad B / K");
    MKB_REQUIRE(2 * K <= 1024, "the pooled path supports size <= 512 (LDS tile list); use the general path");
    MKB_REQUIRE(tb->n_entity <= INT32_MAX, "n_entity too large");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "the pooled path needs head-batch or tail-batch");
    MKB_REQUIRE((((uintptr_t)ws) & 255) == 0, "workspace must be 256-byte aligned");
    return MKB_OK;
}

}  // namespace mkb

using namespace mkb;

extern "C" int64_t mkb_pool_step_workspace_bytes(const mkb_tables_t *tb, int64_t B, int64_t K) {
    if (!tb || B <= 0 || K <= 0) return 0;
    return (int64_t)carve(nullptr, B, 2 * K, tb->entity_dim).bytes;
}

extern "C" int mkb_pool_score_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt,
                                  int64_t B, int64_t K, int mode, float *pool_score, void *ws, void *stream) {
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws)) return rc;
    MKB_REQUIRE(pool_score != nullptr, "pool_score is null");
    const Workspace w = carve(ws, B, 2 * K, tb->entity_dim);
    return dispatch_fwd(tb, mode_is_head(mode), sample, pool, cnt, B, 2 * K, pool_score, w, (hipStream_t)stream);
}

extern "C" int mkb_pool_score_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                                  const uint16_t *cnt, int64_t B, int64_t K, int mode, const float *dpool_score, void *ws,
                                  void *stream) {
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws)) return rc;
    MKB_REQUIRE(gr && gr->g_ent && gr->g_rel && dpool_score, "null pointer");
    MKB_REQUIRE(tb->model != MKB_PROTATE || gr->g_modulus, "pRotatE needs g_modulus");
    Workspace w = carve(ws, B, 2 * K, tb->entity_dim);
    hipStream_t st = (hipStream_t)stream;
    // rebuild the queries (the forward's copy may have been overwritten by another call sharing the workspace)
    RowArgs ra{tb->ent, tb->rel, sample, w.Q, nullptr, nullptr, tb->entity_dim, tb->relation_dim, tb->hidden_dim, tb->phase_div};
    if (int rc = dispatch_query_build(tb, mode_is_head(mode), ra, B, st)) return rc;
    MKB_CHECK_HIP(hipMemcpyAsync(w.G, dpool_score, (size_t)B * 2 * K * 4, hipMemcpyDeviceToDevice, st));
    return dispatch_bwd(tb, mode_is_head(mode), gr, sample, pool, cnt, B, 2 * K, w, st);
}

extern "C" int mkb_pool_step(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                             const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                             float *pos_score, float *pool_score, float *loss, void *ws, void *stream) {
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws)) return rc;
    MKB_REQUIRE(gr && gr->g_ent && gr->g_rel && weight && pos_score && pool_score && loss, "null pointer");
    MKB_REQUIRE(tb->model != MKB_PROTATE || gr->g_modulus, "pRotatE needs g_modulus");
    const int64_t P = 2 * K;
    const Workspace w = carve(ws, B, P, tb->entity_dim);
    hipStream_t st = (hipStream_t)stream;
    const bool head = mode_is_head(mode);
    // positive pass (mode None: tail-style formula against the true tail, pipeline.py:211)
    if (int rc = mkb_score_fwd(tb, sample, nullptr, B, 1, MKB_MODE_DEFAULT, pos_score, stream)) return rc;
    // negative pass over the shared pool (pipeline.py:230-232)
    if (int rc = dispatch_fwd(tb, head, sample, pool, cnt, B, P, pool_score, w, st)) return rc;
    // Adversarial forward + gradient seeds (pipeline.py:234 and the head of :236)
    if (int rc = mkb_adversarial(pos_score, pool_score, weight, cnt, B, P, alpha, loss, w.dpos, w.G, w.scratch, stream)) return rc;
    // backward (pipeline.py:236): pooled negatives, then the positives through the general kernel
    if (int rc = dispatch_bwd(tb, head, gr, sample, pool, cnt, B, P, w, st)) return rc;
    return mkb_score_bwd(tb, gr, sample, nullptr, B, 1, MKB_MODE_DEFAULT, w.dpos, stream);
}

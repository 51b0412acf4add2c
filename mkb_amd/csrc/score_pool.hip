// Pooled scoring kernels: the fast path of the training step (mkb_pool_step / mkb_pool_score_fwd / _bwd).
//
// mkb's sampler draws ONE pool of P = 2K candidate entities per batch and every row filters that same pool
// (sampling/negative_sampling.py:166 is outside the per-row loop at :168).  So the B x K negative block of
// compose/pipeline.py:230-232 is really "B queries x (<= P) shared candidate rows": instead of gathering
// B*K entity rows (2.1 GB at the headline config, models/base.py:193-207) each candidate row is loaded once
// per TILE of 8 batch rows and reused from registers.
//
// All three kernels use 1024-lane workgroups whose lanes OWN the embedding units k (unit = one complex
// number for RotatE, one float otherwise; KPT units per lane so that 1024*KPT covers the row), so a
// workgroup sees whole rows and nothing but the final table gradients ever needs an atomic:
//   pool_fwd    workgroup = (tile of 8 batch rows, slice of the pool positions).  q[8][KPT] in registers;
//               walks the positions used by the tile (compacted list in LDS, wave-uniform "row r uses p" bits
//               -> scalar branches skip unused pairs).  Per position: 8 per-lane partial sums ->
//               v_permlane32_swap / v_permlane16_swap / DPP transposed wave64 reduction (no LDS) ->
//               16 wave totals per row staged in LDS, combined once per batch of positions -> score stored
//               directly (gamma - sum).  No atomics, bit-reproducible.
//   pool_bwd_q  same tiling; dq[8][KPT] accumulates in registers over the slice's positions and is stored
//               once (one partial buffer per slice).  No cross-lane traffic at all.
//   pool_bwd_x  transposed tiling: workgroup = (tile of 8 pool positions, slice of the batch rows);
//               x[8][KPT], dx[8][KPT] in registers, walks the rows that use the tile; dx stored once
//               (one partial buffer per row slice).  The pair term is recomputed instead of exchanging
//               [B,P,D] products through memory or atomics: VALU is cheaper than either here.
//   query_bwd / pool_scatter add the partials and do the only atomics (a few per touched table row).
// VALU-bound by design (RotatE: one v_sqrt / v_rsq per (row, slot, complex dim)).
#include "common.h"
#include "model_math.h"

namespace mkb {

constexpr int kWG = 1024;          // lanes per workgroup (16 waves)
constexpr int kWaves16 = kWG / 64;
constexpr int TI = 8;              // batch rows (fwd / bwd_q) or pool positions (bwd_x) per tile
constexpr int kRing = 4;           // prefetch depth (positions / rows in flight per lane) of the streamed operand
constexpr int kSlab = 16;          // positions per cross-wave reduction batch (forward)
constexpr int kMaxP = 1024;        // pool positions supported by the LDS tile lists
constexpr int kFwdSlices = 2;      // position slices per row tile   (forward)
constexpr int kBwdQSlices = 2;     // position slices per row tile   (backward, dq partial buffers)
constexpr int kBwdXSlices = 8;     // row slices per position tile   (backward, dx partial buffers), minimum
// x-pass occupancy knob (dynamic LDS request).  With 8 row slices and tile-major dispatch the 256 heavy workgroups
// land one per CU and the light tiles co-run beside them; forcing one workgroup per CU (48 KB pad) measured 10 %
// slower (241 vs 219 us for both backward passes), so no pad.
#ifdef MKB_BWDX_PAD
constexpr size_t kOnePerCuPad = 48 * 1024;
#else
constexpr size_t kOnePerCuPad = 0;
#endif

struct PoolArgs {
    const float *ent;      // [N, De]
    const float *Q;        // [B, De] queries
    const int64_t *pool;   // [P]
    const uint16_t *cnt;   // [B, P] multiplicity (0 = row does not use the position)
    const float *G;        // [B, P] d loss / d score (backward)
    float *S;              // [B, P] scores (forward)
    float *dQ;             // [slices, B, De] (backward, q pass)
    float *dX;             // [slices, P, De] (backward, x pass)
    float *g_modulus;      // pRotatE
    const float *modulus;  // pRotatE
    int B, P, d, x_slices;
    int64_t De;
    float kd, c0, c1;      // score = c0 + c1 * sum
};

// exclusive scan of a flag over the 1024-lane workgroup; returns this lane's slot, *total = count
__device__ __forceinline__ int wg_compact_slot(bool flag, int *wave_cnt, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves16; ++w) {
        const int c = wave_cnt[w];
        off += (w < wave) ? c : 0;
        tot += c;
    }
    *total = tot;
    return off + __popcll(b & ((1ull << lane) - 1ull));
}

// 8 per-lane values -> wave totals via half-wave / row swaps and DPP (no LDS).  On return every lane of
// 16-lane row R (= lane >> 4) holds t0 = total of value 4*(R>>1) + 2*(R&1) and t1 = total of that + 1.
__device__ __forceinline__ void reduce8_wave(const float (&v)[8], float &t0, float &t1) {
    float w[4], u[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // lanes 0-31 keep values 0-3, lanes 32-63 keep values 4-7
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 4]), false, false);
        w[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // even 16-lane rows keep j, odd rows keep j + 2
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[j]), __float_as_uint(w[j + 2]), false, false);
        u[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // all-reduce inside the 16-lane row: ror 8, half-mirror, two quad perms
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

// ------------------------------------------------------------------------------------------------ forward
template <int MODEL, bool HEAD, int KPT>
__global__ __launch_bounds__(kWG) void pool_fwd_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    __shared__ int s_row[kMaxP];                      // entity id per active position
    __shared__ int s_pos[kMaxP];                      // pool position
    __shared__ unsigned s_mask[kMaxP];                // bit r: row r of the tile uses it
    __shared__ float s_part[2][kSlab][kWaves16][TI];  // wave totals, double buffered
    __shared__ int s_wave_cnt[kWaves16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;

    // positions used by at least one row of the tile (P <= 1024: one position per lane)
    unsigned m_own = 0;
    if (tid < A.P) {
        unsigned c[TI];
#pragma unroll
        for (int r = 0; r < TI; ++r) c[r] = (i0 + r < A.B) ? A.cnt[(int64_t)(i0 + r) * A.P + tid] : 0;
#pragma unroll
        for (int r = 0; r < TI; ++r) m_own |= (c[r] != 0) ? (1u << r) : 0u;
    }
    int n_act;
    const int slot = wg_compact_slot(m_own != 0, s_wave_cnt, &n_act);
    if (m_own != 0) {
        s_pos[slot] = tid;
        s_mask[slot] = m_own;
        s_row[slot] = (int)A.pool[tid];
    }
    __syncthreads();

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

    // this workgroup's slice: active positions a = slice, slice + nslices, ...
    const int nsl = gridDim.y, sl = blockIdx.y;
    const int n_mine = (n_act > sl) ? (n_act - sl + nsl - 1) / nsl : 0;

    // Candidate rows stream through a kRing-deep register ring: the first touch of a pool row by an XCD comes from
    // Infinity Cache / HBM (~1 us), several positions' worth of compute, so one-ahead prefetch left every wave
    // waiting on it (measured: loads cost 55 of 124 us).
    float xr0[kRing][KPT], xr1[kRing][KPT];
    auto load_x = [&](int j, float (&d0)[KPT], float (&d1)[KPT]) {
        const float *x = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(s_row[sl + j * nsl]) * A.De;
#pragma unroll
        for (int v = 0; v < KPT; ++v) {  // unconditional loads from a clamped address + select: a predicated load
            const bool ok = u0 + v < NU;  // becomes a branch whose join makes the compiler wait for it at once
            const int uu = min(u0 + v, NU - 1);
            const float l0 = x[uu];
            const float l1 = CP ? x[A.d + uu] : 0.f;
            d0[v] = ok ? l0 : 0.f;
            d1[v] = ok ? l1 : 0.f;
        }
    };
    // Loads are UNCONDITIONAL (index clamped to the last position; the tail re-loads it) so that the compiler can
    // count outstanding loads and wait with vmcnt(N) for the oldest only; a load under a branch forces vmcnt(0).
    const int j_last = n_mine - 1;
    if (n_mine > 0) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) load_x(min(s, j_last), xr0[s], xr1[s]);
    }
    for (int jb = 0; jb < n_mine; jb += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int j = jb + s;
            {
                const int jj = j % kSlab, buf = (j / kSlab) & 1;
                const unsigned m = (j < n_mine) ? __builtin_amdgcn_readfirstlane(s_mask[sl + min(j, j_last) * nsl]) : 0u;
                float x0[KPT], x1[KPT];
#pragma unroll
                for (int v = 0; v < KPT; ++v) { x0[v] = xr0[s][v]; x1[v] = xr1[s][v]; }
#ifndef MKB_ABL_NOLOAD
                load_x(min(j + kRing, j_last), xr0[s], xr1[s]);
#endif
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
                float t0, t1;
#ifdef MKB_ABL_FWD_NOREDUCE
                t0 = part[0] + part[1] + part[2] + part[3]; t1 = part[4] + part[5] + part[6] + part[7];
#else
                reduce8_wave(part, t0, t1);
#endif
                if ((lane & 15) == 0) {
                    const int R = lane >> 4, r = 4 * (R >> 1) + 2 * (R & 1);
                    s_part[buf][jj][wave][r] = t0;
                    s_part[buf][jj][wave][r + 1] = t1;
                }
                if (j < n_mine && (jj == kSlab - 1 || j == j_last)) {  // wave-uniform: combine <= kSlab positions
                    __syncthreads();  // double-buffered s_part: one barrier per batch
                    const int j0 = j - jj, nb = jj + 1;
                    if (tid < nb * TI) {
                        const int cj = tid / TI, r = tid % TI;
                        const int a = sl + (j0 + cj) * nsl;
                        if (s_mask[a] & (1u << r)) {
                            float sum = 0.f;
#pragma unroll
                            for (int w = 0; w < kWaves16; ++w) sum += s_part[buf][cj][w][r];
                            if constexpr (MODEL == MKB_PROTATE) sum *= A.modulus[0];  // gamma - modulus * sum (protate.py:91)
                            A.S[(int64_t)(i0 + r) * A.P + s_pos[a]] = A.c0 + A.c1 * sum;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dq
template <int MODEL, bool HEAD, int KPT>
__global__ __launch_bounds__(kWG) void pool_bwd_q_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    __shared__ int s_row[kMaxP];
    __shared__ unsigned s_mask[kMaxP];
    __shared__ __attribute__((aligned(16))) float s_g[kMaxP][TI];  // gradient seeds of the tile per active position
    __shared__ int s_wave_cnt[kWaves16];
    __shared__ float s_red[kWaves16];

    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;

    unsigned m_own = 0;
    float g_own[TI];
    if (tid < A.P) {
        unsigned c[TI];
#pragma unroll
        for (int r = 0; r < TI; ++r) {
            const bool in = i0 + r < A.B;
            c[r] = in ? A.cnt[(int64_t)(i0 + r) * A.P + tid] : 0;
            g_own[r] = in ? A.G[(int64_t)(i0 + r) * A.P + tid] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < TI; ++r) m_own |= (c[r] != 0) ? (1u << r) : 0u;
    }
    int n_act;
    const int slot = wg_compact_slot(m_own != 0, s_wave_cnt, &n_act);
    if (m_own != 0) {
        s_mask[slot] = m_own;
        s_row[slot] = (int)A.pool[tid];
#pragma unroll
        for (int r = 0; r < TI; ++r) s_g[slot][r] = g_own[r];
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

    const int nsl = gridDim.y, sl = blockIdx.y;
    const int n_mine = (n_act > sl) ? (n_act - sl + nsl - 1) / nsl : 0;
    float xr0[kRing][KPT], xr1[kRing][KPT];
    auto load_x = [&](int j, float (&d0)[KPT], float (&d1)[KPT]) {
        const float *x = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(s_row[sl + j * nsl]) * A.De;
#pragma unroll
        for (int v = 0; v < KPT; ++v) {  // unconditional loads from a clamped address + select: a predicated load
            const bool ok = u0 + v < NU;  // becomes a branch whose join makes the compiler wait for it at once
            const int uu = min(u0 + v, NU - 1);
            const float l0 = x[uu];
            const float l1 = CP ? x[A.d + uu] : 0.f;
            d0[v] = ok ? l0 : 0.f;
            d1[v] = ok ? l1 : 0.f;
        }
    };
    const int j_last = n_mine - 1;
    if (n_mine > 0) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) load_x(min(s, j_last), xr0[s], xr1[s]);
    }
    for (int jb = 0; jb < n_mine; jb += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int j = jb + s;
            {
                const int a = sl + min(j, j_last) * nsl;
                const unsigned m = (j < n_mine) ? __builtin_amdgcn_readfirstlane(s_mask[a]) : 0u;
                float g[TI];
#pragma unroll
                for (int r = 0; r < TI; ++r) g[r] = s_g[a][r];  // two ds_read_b128, same address in every lane
                float x0[KPT], x1[KPT];
#pragma unroll
                for (int v = 0; v < KPT; ++v) { x0[v] = xr0[s][v]; x1[v] = xr1[s][v]; }
#ifndef MKB_ABL_NOLOAD
                load_x(min(j + kRing, j_last), xr0[s], xr1[s]);
#endif
#pragma unroll
                for (int r = 0; r < TI; ++r) {
                    if (m & (1u << r)) {
#pragma unroll
                        for (int v = 0; v < KPT; ++v) {
                            if constexpr (CP) {
                                Cplx dq, dx;
                                pair_bwd_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]}, g[r], dq, dx);
                                dq0[r][v] += dq.re;
                                dq1[r][v] += dq.im;
                            } else {
                                float dq, dx, e0 = 0.f;
                                pair_bwd_real<MODEL, HEAD>(q0[r][v], x0[v], g[r], A.kd, modulus, dq, dx, e0);
                                dq0[r][v] += dq;
                                extra += g[r] * e0;
                            }
                        }
                    }
                }
            }
        }
    }
    float *dQs = A.dQ + (int64_t)sl * A.B * A.De;
#pragma unroll
    for (int r = 0; r < TI; ++r)
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if ((i0 + r < A.B) && (u0 + v < NU)) {
                float *dqrow = dQs + (int64_t)(i0 + r) * A.De;
                dqrow[u0 + v] = dq0[r][v];
                if constexpr (CP) dqrow[A.d + u0 + v] = dq1[r][v];
            }
        }
    if constexpr (MODEL == MKB_PROTATE) {  // d score / d modulus = - sum_k |sin z|   (protate.py:91)
        extra = wave_sum(extra);
        if ((tid & 63) == 0) s_red[tid >> 6] = extra;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < kWaves16; ++w) s += s_red[w];
            atomicAdd(A.g_modulus, -s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dx
template <int MODEL, bool HEAD, int KPT>
__global__ __launch_bounds__(kWG) void pool_bwd_x_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    __shared__ int s_i[kWG];          // batch rows of the slice that use the position tile
    __shared__ unsigned s_mask[kWG];  // bit t: the row uses position p0 + t
    __shared__ __attribute__((aligned(16))) float s_g[kWG][TI];
    __shared__ int s_wave_cnt[kWaves16];

    const int tid = threadIdx.x;
    // 1-D grid, tile-major: the workgroups of the low position tiles (used by every row: the heavy ones) are
    // dispatched first, one per CU; the light / empty tiles fill in behind them.
    const int nsl = A.x_slices, sl = blockIdx.x % nsl;
    const int p0 = (blockIdx.x / nsl) * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;
    const int rows_per = (A.B + nsl - 1) / nsl;  // <= 1024 (host picks nsl)
    const int i_own = sl * rows_per + tid;

    unsigned m_own = 0;
    float g_own[TI];
#pragma unroll
    for (int t = 0; t < TI; ++t) g_own[t] = 0.f;
    if (tid < rows_per && i_own < A.B) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            if (p0 + t < A.P) {
                const unsigned c = A.cnt[(int64_t)i_own * A.P + p0 + t];
                g_own[t] = A.G[(int64_t)i_own * A.P + p0 + t];
                m_own |= (c != 0) ? (1u << t) : 0u;
            }
        }
    }
    int n_rows;
    const int slot = wg_compact_slot(m_own != 0, s_wave_cnt, &n_rows);
    if (m_own != 0) {
        s_i[slot] = i_own;
        s_mask[slot] = m_own;
#pragma unroll
        for (int t = 0; t < TI; ++t) s_g[slot][t] = g_own[t];
    }
    __syncthreads();

    float x0[TI][KPT], x1[TI][KPT], dx0[TI][KPT], dx1[TI][KPT];
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const bool pin = p0 + t < A.P;
        const float *x = A.ent + (pin ? A.pool[p0 + t] : 0) * A.De;
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = pin && (u0 + v < NU) && n_rows > 0;
            x0[t][v] = ok ? x[u0 + v] : 0.f;
            x1[t][v] = (CP && ok) ? x[A.d + u0 + v] : 0.f;
            dx0[t][v] = 0.f;
            dx1[t][v] = 0.f;
        }
    }
    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;

    float qr0[kRing][KPT], qr1[kRing][KPT];
    auto load_q = [&](int j, float (&d0)[KPT], float (&d1)[KPT]) {
        const float *q = A.Q + (int64_t)__builtin_amdgcn_readfirstlane(s_i[j]) * A.De;
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            const bool ok = u0 + v < NU;
            const int uu = min(u0 + v, NU - 1);
            const float l0 = q[uu];
            const float l1 = CP ? q[A.d + uu] : 0.f;
            d0[v] = ok ? l0 : 0.f;
            d1[v] = ok ? l1 : 0.f;
        }
    };
    const int j_last = n_rows - 1;
    if (n_rows > 0) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) load_q(min(s, j_last), qr0[s], qr1[s]);
    }
    for (int jb = 0; jb < n_rows; jb += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int j = jb + s;
            {
                const int jc = min(j, j_last);
                const unsigned m = (j < n_rows) ? __builtin_amdgcn_readfirstlane(s_mask[jc]) : 0u;
                float g[TI];
#pragma unroll
                for (int t = 0; t < TI; ++t) g[t] = s_g[jc][t];
                float q0[KPT], q1[KPT];
#pragma unroll
                for (int v = 0; v < KPT; ++v) { q0[v] = qr0[s][v]; q1[v] = qr1[s][v]; }
#ifndef MKB_ABL_NOLOAD
                load_q(min(j + kRing, j_last), qr0[s], qr1[s]);
#endif
#pragma unroll
                for (int t = 0; t < TI; ++t) {
                    if (m & (1u << t)) {
#pragma unroll
                        for (int v = 0; v < KPT; ++v) {
                            if constexpr (CP) {
                                Cplx dq, dx;
                                pair_bwd_cmod(Cplx{q0[v], q1[v]}, Cplx{x0[t][v], x1[t][v]}, g[t], dq, dx);
                                dx0[t][v] += dx.re;
                                dx1[t][v] += dx.im;
                            } else {
                                float dq, dx, e0 = 0.f;
                                pair_bwd_real<MODEL, HEAD>(q0[v], x0[t][v], g[t], A.kd, modulus, dq, dx, e0);
                                dx0[t][v] += dx;
                            }
                        }
                    }
                }
            }
        }
    }
    float *dXs = A.dX + (int64_t)sl * A.P * A.De;
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if ((p0 + t < A.P) && (u0 + v < NU)) {
                float *row = dXs + (int64_t)(p0 + t) * A.De;
                row[u0 + v] = dx0[t][v];
                if constexpr (CP) row[A.d + u0 + v] = dx1[t][v];
            }
        }
}

// ------------------------------------------------------------------------------------------------ row kernels
struct RowArgs {
    const float *ent, *rel;
    const int64_t *sample;
    float *Q;          // [B, De] out (build) / [nslices, B, De] dQ partials in (backward)
    float *g_ent, *g_rel;
    int64_t De, Dr;
    int d, B, nslices;
    float kd;
};

// Q[i] = query of row i (same math as the LDS staging of the general forward kernel, written to global)
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

// chain dQ[i] (sum of the slice partials) into the fixed operands' gradient rows (duplicate rows add: atomics)
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void query_bwd_kernel(RowArgs A) {
    const int64_t i = blockIdx.x;
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];
    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    const float *dq = A.Q + i * A.De;
    const int64_t sstride = (int64_t)A.B * A.De;
    float *g_e = A.g_ent + (HEAD ? t : h) * A.De;
    float *g_r = A.g_rel + r * A.Dr;
    auto dq_at = [&](int k) {
        float s = 0.f;
        for (int sl = 0; sl < A.nslices; ++sl) s += dq[sl * sstride + k];
        return s;
    };
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        const float *e = HEAD ? et : eh;
        for (int u = threadIdx.x; u < A.d; u += 256) {
            Cplx de, dr;
            query_bwd_cplx<MODEL, HEAD>(Cplx{e[u], e[A.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f},
                                        Cplx{dq_at(u), dq_at(A.d + u)}, A.kd, de, dr);
            atomicAdd(g_e + u, de.re);
            atomicAdd(g_e + A.d + u, de.im);
            atomicAdd(g_r + u, dr.re);
            if constexpr (MODEL == MKB_COMPLEX) atomicAdd(g_r + A.d + u, dr.im);
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256) {
            float da, db;
            query_bwd_real<MODEL, HEAD>(HEAD ? er[u] : eh[u], HEAD ? et[u] : er[u], dq_at(u), A.kd, da, db);
            atomicAdd((HEAD ? g_r : g_e) + u, da);
            atomicAdd((HEAD ? g_e : g_r) + u, db);
        }
    }
}

// g_ent[pool[p]] += sum over slices of dX[slice][p]   (pool positions may repeat an entity: atomics)
__global__ __launch_bounds__(256) void pool_scatter_kernel(const float *__restrict__ dX, int nslices, int P,
                                                           const int64_t *__restrict__ pool, float *__restrict__ g_ent,
                                                           int64_t De) {
    const int64_t p = blockIdx.x;
    float *dst = g_ent + pool[p] * De;
    for (int64_t k = threadIdx.x; k < De; k += 256) {
        float v = 0.f;
        for (int sl = 0; sl < nslices; ++sl) v += dX[((int64_t)sl * P + p) * De + k];
        if (v != 0.f) atomicAdd(dst + k, v);
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct Workspace {
    float *Q, *dQ, *G, *dX, *dpos, *scratch;
    int x_slices;
    size_t bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int x_slices_for(int64_t B) {
    int64_t n = (B + kWG - 1) / kWG;
    return (int)(n > kBwdXSlices ? n : kBwdXSlices);
}

static Workspace carve(void *ws, int64_t B, int64_t P, int64_t De) {
    Workspace w;
    unsigned char *p = (unsigned char *)ws;
    size_t off = 0;
    auto take = [&](size_t n) { void *r = p ? p + off : nullptr; off += align256(n); return (float *)r; };
    w.x_slices = x_slices_for(B);
    w.Q = take((size_t)B * De * 4);
    w.dQ = take((size_t)kBwdQSlices * B * De * 4);
    w.G = take((size_t)B * P * 4);
    w.dX = take((size_t)w.x_slices * P * De * 4);
    w.dpos = take((size_t)B * 4);
    w.scratch = take((size_t)(B + 1) * 4);
    w.bytes = off;
    return w;
}

static int units_of(const mkb_tables_t *tb) { return tb->model == MKB_ROTATE ? tb->hidden_dim : (int)tb->entity_dim; }

static PoolArgs make_args(const mkb_tables_t *tb, const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t P,
                          const Workspace &w) {
    PoolArgs A{};
    A.ent = tb->ent; A.Q = w.Q; A.pool = pool; A.cnt = cnt; A.G = w.G; A.dQ = w.dQ; A.dX = w.dX;
    A.B = (int)B; A.P = (int)P; A.d = tb->hidden_dim; A.De = tb->entity_dim; A.kd = tb->phase_div;
    A.modulus = tb->modulus;
    return A;
}

template <int MODEL, bool HEAD>
static int run_query_build(const RowArgs &ra, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL((query_build_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int run_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                   int64_t P, float *S, const Workspace &w, hipStream_t st) {
    RowArgs ra{tb->ent, tb->rel, sample, w.Q, nullptr, nullptr, tb->entity_dim, tb->relation_dim, tb->hidden_dim, (int)B, 1,
               tb->phase_div};
    if (int rc = run_query_build<MODEL, HEAD>(ra, B, st)) return rc;
    MKB_CHECK_HIP(hipMemsetAsync(S, 0, (size_t)B * P * 4, st));
    PoolArgs A = make_args(tb, pool, cnt, B, P, w);
    A.S = S;
    A.c0 = ModelTraits<MODEL>::uses_gamma ? tb->gamma : 0.f;
    A.c1 = ModelTraits<MODEL>::uses_gamma ? -1.f : 1.f;
    dim3 grid((unsigned)((B + TI - 1) / TI), kFwdSlices);
    const int NU = units_of(tb);
    {
        ProfScope ps(MKB_PROF_POOL_FWD, st);
        if (NU <= kWG) hipLaunchKernelGGL((pool_fwd_kernel<MODEL, HEAD, 1>), grid, dim3(kWG), 0, st, A);
        else if (NU <= 2 * kWG) hipLaunchKernelGGL((pool_fwd_kernel<MODEL, HEAD, 2>), grid, dim3(kWG), 0, st, A);
        else hipLaunchKernelGGL((pool_fwd_kernel<MODEL, HEAD, 4>), grid, dim3(kWG), 0, st, A);
    }
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int run_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                   const uint16_t *cnt, int64_t B, int64_t P, const Workspace &w, hipStream_t st) {
    PoolArgs A = make_args(tb, pool, cnt, B, P, w);
    A.g_modulus = gr->g_modulus;
    const int NU = units_of(tb);
    dim3 gq((unsigned)((B + TI - 1) / TI), kBwdQSlices);
    dim3 gx((unsigned)(((P + TI - 1) / TI) * w.x_slices));
    A.x_slices = w.x_slices;
    {
        ProfScope ps(MKB_PROF_POOL_BWD, st);
        if (NU <= kWG) {
            hipLaunchKernelGGL((pool_bwd_q_kernel<MODEL, HEAD, 1>), gq, dim3(kWG), 0, st, A);
            hipLaunchKernelGGL((pool_bwd_x_kernel<MODEL, HEAD, 1>), gx, dim3(kWG), kOnePerCuPad, st, A);
        } else if (NU <= 2 * kWG) {
            hipLaunchKernelGGL((pool_bwd_q_kernel<MODEL, HEAD, 2>), gq, dim3(kWG), 0, st, A);
            hipLaunchKernelGGL((pool_bwd_x_kernel<MODEL, HEAD, 2>), gx, dim3(kWG), kOnePerCuPad, st, A);
        } else {
            hipLaunchKernelGGL((pool_bwd_q_kernel<MODEL, HEAD, 4>), gq, dim3(kWG), 0, st, A);
            hipLaunchKernelGGL((pool_bwd_x_kernel<MODEL, HEAD, 4>), gx, dim3(kWG), kOnePerCuPad, st, A);
        }
    }
    RowArgs ra{tb->ent, tb->rel, sample, w.dQ, gr->g_ent, gr->g_rel, tb->entity_dim, tb->relation_dim, tb->hidden_dim,
               (int)B, kBwdQSlices, tb->phase_div};
    hipLaunchKernelGGL((query_bwd_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    hipLaunchKernelGGL(pool_scatter_kernel, dim3((unsigned)P), dim3(256), 0, st, w.dX, w.x_slices, (int)P, pool, gr->g_ent,
                       tb->entity_dim);
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
static int dispatch_query_build(const mkb_tables_t *tb, bool head, const RowArgs &ra, int64_t B, hipStream_t st) {
    MKB_DISPATCH(run_query_build, tb->model, head, ra, B, st);
}

static int check_pool_call(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                           int64_t K, int mode, const void *ws) {
    if (int rc = validate_tables(tb)) return rc;
    MKB_REQUIRE(sample && pool && cnt && ws, "null pointer");
    MKB_REQUIRE(B > 0 && K > 0 && B <= INT32_MAX, "bad B / K");
    MKB_REQUIRE(2 * K <= kMaxP, "the pooled path supports size <= 512 (LDS tile lists); use the general path");
    MKB_REQUIRE(units_of(tb) <= 4 * kWG, "the pooled path supports rows of <= 4096 units; use the general path");
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
    RowArgs ra{tb->ent, tb->rel, sample, w.Q, nullptr, nullptr, tb->entity_dim, tb->relation_dim, tb->hidden_dim, (int)B, 1,
               tb->phase_div};
    if (int rc = dispatch_query_build(tb, mode_is_head(mode), ra, B, st)) return rc;
    MKB_CHECK_HIP(hipMemcpyAsync(w.G, dpool_score, (size_t)B * 2 * K * 4, hipMemcpyDeviceToDevice, st));
    return dispatch_bwd(tb, mode_is_head(mode), gr, sample, pool, cnt, B, 2 * K, w, st);
}

extern "C" int mkb_pool_step(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                             const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                             const float *weight_sum, float *pos_score, float *pool_score, float *loss, void *ws,
                             void *stream) {
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
    if (int rc = mkb_adversarial(pos_score, pool_score, weight, cnt, B, P, alpha, weight_sum, loss, w.dpos, w.G, w.scratch, stream)) return rc;
    // backward (pipeline.py:236): pooled negatives, then the positives through the general kernel
    if (int rc = dispatch_bwd(tb, head, gr, sample, pool, cnt, B, P, w, st)) return rc;
    return mkb_score_bwd(tb, gr, sample, nullptr, B, 1, MKB_MODE_DEFAULT, w.dpos, stream);
}

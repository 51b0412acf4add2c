// Pooled scoring kernels (templates).  Instantiated per model in score_pool_<model>.hip, driven by score_pool.hip.
//
// mkb's sampler draws ONE pool of P = 2K candidate entities per batch and every row filters that same pool
// (sampling/negative_sampling.py:166 is outside the per-row loop at :168).  So the B x K negative block of
// compose/pipeline.py:230-232 is really "B queries x (<= P) shared candidate rows": instead of gathering
// B*K entity rows (2.1 GB at the headline config, models/base.py:193-207) each candidate row is loaded once
// per TILE of 8 batch rows and reused from registers.
//
// All three kernels use workgroups of NW waves whose lanes OWN the embedding units k (unit = one complex number
// for RotatE, one float otherwise; KPT consecutive units per lane, NW*64*KPT >= units per row), so a workgroup
// sees whole rows and nothing but the final table gradients ever needs an atomic:
//   pool_fwd    workgroup = (tile of 8 batch rows, slice of the pool positions).  q[8][KPT] in registers;
//               walks the positions used by the tile (compacted list in LDS, wave-uniform "row r uses p" bits
//               -> scalar branches skip unused pairs).  Per position: 8 per-lane partial sums ->
//               v_permlane32_swap / v_permlane16_swap / DPP transposed wave64 reduction (no LDS) ->
//               NW wave totals per row staged in LDS, combined once per batch of 16 positions -> score stored
//               directly (gamma - sum).  No atomics, bit-reproducible.
//   pool_bwd_q  same tiling; dq[8][KPT] accumulates in registers over the slice's positions and is stored
//               once (one partial buffer per slice).  No cross-lane traffic at all.
//   pool_bwd_x  transposed tiling: workgroup = (tile of 8 pool positions, slice of the batch rows);
//               x[8][KPT], dx[8][KPT] in registers, walks the rows that use the tile; dx is added to the table
//               gradient row once per workgroup (8 atomics per element and step).  The pair term is recomputed
//               instead of exchanging [B,P,D] products through memory or per-pair atomics: VALU is cheaper.
// KPT = 2 / 4 lanes load float2 / float4 and give the compiler pairs of units to pack into v_pk_* ops.
// VALU-bound by design (RotatE: one v_sqrt / v_rsq per (row, slot, complex dim)).
#pragma once
#include <type_traits>
#include "common.h"
#include "model_math.h"

namespace mkb {

constexpr int TI = 8;       // batch rows (fwd / bwd_q) or pool positions (bwd_x) per tile
constexpr int kRing = 4;    // prefetch depth of the streamed operand (positions / rows in flight per lane)
constexpr int kRing1 = 3;   // candidate rows in flight per lane in the single-pass backward
constexpr int kRingF = 1;   // ... of the fringe's dq half when the dense pass is on (few positions per wave; every unrolled copy is cold code)
constexpr int kDense = 6;   // a streamed item used by >= kDense of the tile's 8 rows / positions takes the branch-free body
constexpr int kSlab = 16;   // positions per cross-wave reduction batch (forward)
constexpr int kMaxP = 2048; // pool positions supported (= the device sampler's limit, size <= 1024)
constexpr int kMaxSlices = 8;
constexpr int kMfmaDqSlices = 2;  // MFMA route: dQ partial products a K split of G . X may leave for the row backward to add up

struct DxReduce;

struct PoolArgs {
    const float *ent;      // [N, De]
    const float *Q;        // [B, De] queries
    const int64_t *pool;   // [P]
    const uint16_t *cnt;   // [B, P] multiplicity (0 = row does not use the position)
    const float *G;        // [B, P] d loss / d score (backward)
    float *S;              // [B, P] scores (forward)
    float *dQ;             // [slices, B, De] (backward, q pass)
    float *g_ent;          // [N, De] table gradient (backward, x pass adds into it)
    float *g_modulus;      // pRotatE
    const float *modulus;  // pRotatE
    int B, P, d, x_slices, q_slices;  // q_slices: dQ partial buffers (= position blocks of the single-pass backward)
    int p_lo;                         // forward: first pool position this launch's sparse workgroups cover (0 = the whole pool)
    float *tile_part;                 // forward tile (host side): partial-sum buffer and where to describe the pending reduction
    GemmTail *tile_tail;
    int x_blocks, q_first; // merged backward launch: q_first dq blocks, then x_blocks dx blocks, then the other dq blocks
    int dim_slices, pb_halves, tiles_per_wave;  // single-pass backward (pool_bwd1_kernel)
    int dense_lanes;             // ... lanes [0, dense_lanes) of every half hold positions of the dense prefix (0 = no dense pass)
    int lds_ids_off;             // ... dense pass: offset (ints, behind the header) of the block's pool-id table in LDS
    float *dXp;                  // [row groups][blocks][slots][dim slices][64][NC] dx partials of the single-pass backward
    unsigned long long *xused;   // [row groups][blocks][8] used-slot masks of each (row group, block)
    DxReduce *dx_reduce_out;     // host side: non-null = do not launch the reduction, describe it here instead
    int g_blocked;               // G is in the tile-blocked seed layout of common.h (SeedLayout of this launch's blocks / halves)
    int *occ;                    // forward inside mkb_pool_step: count the batch's entities (score_pool.hip RowStepArgs::occ)
    const int64_t *occ_sample;   //   ... heads and tails of sample [B, 3], and the pool ids
    int64_t De;
    float kd, c0, c1;      // score = c0 + c1 * sum
#ifdef MKB_TRACE_WG
    unsigned long long *trace;  // tools/wgtrace.py: 8 words per workgroup (timestamps, hardware id, work)
    int trace_kind;             // which kernel records: 0 fwd, 1 bwd_q, 2 bwd_x
#endif
};

#ifdef MKB_TRACE_WG
#define MKB_TRACE_T(var) const unsigned long long var = wall_clock64()
#define MKB_TRACE_OUT(A, kind, t0, t1, t2, work)                                                    \
    if (threadIdx.x == 0 && (A).trace && ((A).trace_kind == (kind) || (A).trace_kind == 3)) {                                                            \
        unsigned long long *tr = (A).trace + ((A).trace_kind == 3 ? 8ull * 4096 * (kind) : 0ull) + 8ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y); \
        unsigned hw = 0, xcc = 0;                                                                   \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                           \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                         \
        tr[0] = t0; tr[1] = t1; tr[2] = t2; tr[3] = wall_clock64();                                 \
        tr[4] = hw; tr[5] = xcc; tr[6] = (unsigned long long)(work); tr[7] = blockIdx.x;            \
    }
#define MKB_TRACE_ONLY(...) __VA_ARGS__
#else
#define MKB_TRACE_T(var)
#define MKB_TRACE_OUT(A, kind, t0, t1, t2, work)
#define MKB_TRACE_ONLY(...)
#endif

// a value every lane of the wave holds identically -> scalar register
__device__ __forceinline__ float uniform_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// exclusive scan of a flag over the NW-wave workgroup; returns this lane's slot, *total = count.
// Contains one barrier; callers put another one before the next call (wave_cnt is reused).
template <int NW>
__device__ __forceinline__ int wg_compact_slot(bool flag, int *wave_cnt, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    if (lane == 0) wave_cnt[wave] = __popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
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

template <int N> struct AccVecOf;
template <> struct AccVecOf<1> { typedef float type; };
template <> struct AccVecOf<2> { typedef float2 type; };
template <> struct AccVecOf<4> { typedef float4 type; };

// Load this lane's KPT consecutive units of a row: UNCONDITIONAL vector load from a clamped offset, then a select.
// (A predicated load becomes a branch whose join makes the compiler wait for the data immediately, which
// defeats the prefetch ring.)  KPT > 1 requires the row halves to be KPT*4-byte aligned (host checks).
template <bool CP, int KPT>
__device__ __forceinline__ void load_units(const float *__restrict__ row, int d, int NU, int u0, float (&d0)[KPT],
                                           float (&d1)[KPT]) {
    const bool ok = u0 < NU;  // NU is a multiple of KPT when KPT > 1: all of a lane's units are in or out together
    const int uu = ok ? u0 : 0;
    if constexpr (KPT == 1) {
        const float a = row[uu];
        const float b = CP ? row[d + uu] : 0.f;
        d0[0] = ok ? a : 0.f;
        d1[0] = ok ? b : 0.f;
    } else if constexpr (KPT == 2) {
        const float2 a = *reinterpret_cast<const float2 *>(row + uu);
        const float2 b = CP ? *reinterpret_cast<const float2 *>(row + d + uu) : make_float2(0.f, 0.f);
        d0[0] = ok ? a.x : 0.f; d0[1] = ok ? a.y : 0.f;
        d1[0] = ok ? b.x : 0.f; d1[1] = ok ? b.y : 0.f;
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(row + uu);
        const float4 b = CP ? *reinterpret_cast<const float4 *>(row + d + uu) : make_float4(0.f, 0.f, 0.f, 0.f);
        d0[0] = ok ? a.x : 0.f; d0[1] = ok ? a.y : 0.f; d0[2] = ok ? a.z : 0.f; d0[3] = ok ? a.w : 0.f;
        d1[0] = ok ? b.x : 0.f; d1[1] = ok ? b.y : 0.f; d1[2] = ok ? b.z : 0.f; d1[3] = ok ? b.w : 0.f;
    }
}

// The same load without the select: lanes past the row's end hold data of unit 0 until mask_units() zeroes them.  (Selecting
// at the load makes the wave wait for the data there and then; the dense pass of the single-pass backward requests a row one
// position ahead and masks it when it takes it over.)
template <bool CP, int KPT>
__device__ __forceinline__ void load_units_raw(const float *__restrict__ row, int d, int NU, int u0, float (&d0)[KPT],
                                               float (&d1)[KPT]) {
    const int uu = u0 < NU ? u0 : 0;
    typedef typename AccVecOf<KPT>::type vec_t;
    const vec_t a = *reinterpret_cast<const vec_t *>(row + uu);
    vec_t b = a;
    if constexpr (CP) b = *reinterpret_cast<const vec_t *>(row + d + uu);
    if constexpr (KPT == 1) { d0[0] = a; d1[0] = CP ? b : 0.f; }
    else if constexpr (KPT == 2) { d0[0] = a.x; d0[1] = a.y; d1[0] = CP ? b.x : 0.f; d1[1] = CP ? b.y : 0.f; }
    else {
        d0[0] = a.x; d0[1] = a.y; d0[2] = a.z; d0[3] = a.w;
        d1[0] = CP ? b.x : 0.f; d1[1] = CP ? b.y : 0.f; d1[2] = CP ? b.z : 0.f; d1[3] = CP ? b.w : 0.f;
    }
}

template <bool CP, int KPT>
__device__ __forceinline__ void store_units(float *__restrict__ row, int d, int NU, int u0, const float (&s0)[KPT],
                                            const float (&s1)[KPT]) {
    if (u0 >= NU) return;
    if constexpr (KPT == 1) {
        row[u0] = s0[0];
        if constexpr (CP) row[d + u0] = s1[0];
    } else if constexpr (KPT == 2) {
        *reinterpret_cast<float2 *>(row + u0) = make_float2(s0[0], s0[1]);
        if constexpr (CP) *reinterpret_cast<float2 *>(row + d + u0) = make_float2(s1[0], s1[1]);
    } else {
        *reinterpret_cast<float4 *>(row + u0) = make_float4(s0[0], s0[1], s0[2], s0[3]);
        if constexpr (CP) *reinterpret_cast<float4 *>(row + d + u0) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    }
}

// ------------------------------------------------------------------------------------------------ forward
// Body of one (row tile bx of n_bx, position slice sl of nsl) workgroup over the pool positions [A.p_lo, A.P) -- the whole
// pool for pool_fwd_kernel, the sparse fringe behind the dense prefix when it rides pool_fwd_tile_kernel's launch.
template <int MODEL, bool HEAD, int KPT, int NW>
__device__ __forceinline__ void pool_fwd_body(const PoolArgs &A, const int bx, const int n_bx, const int sl, const int nsl, int *lds_dyn) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int WG = NW * 64;
    // position slice of this workgroup: p = p_lo + slice + nslices * i (interleaved, so every slice gets the same share of the
    // low, heavily used positions); only those are listed: LDS (and the list-building work) shrink with the slices
    const int Pw = A.P - A.p_lo;
    const int Pn = (Pw - sl + nsl - 1) / nsl;
    const int Pcap = (Pw + nsl - 1) / nsl;
    int *s_row = lds_dyn;                                           // entity id per active position
    int *s_pos = lds_dyn + Pcap;                                    // pool position
    unsigned *s_mask = reinterpret_cast<unsigned *>(lds_dyn + 2 * Pcap);  // bit r: row r of the tile uses it
    __shared__ float s_part[2][kSlab][NW][TI];  // wave totals, double buffered
    __shared__ int s_wave_cnt[NW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = bx * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;

    // the tile's query rows: issued first so that their latency overlaps the list build below
    float q0[TI][KPT], q1[TI][KPT];
#pragma unroll
    for (int r = 0; r < TI; ++r) {
        load_units<CP, KPT>(A.Q + (int64_t)min(i0 + r, A.B - 1) * A.De, A.d, NU, u0, q0[r], q1[r]);
        if (i0 + r >= A.B) {
#pragma unroll
            for (int v = 0; v < KPT; ++v) { q0[r][v] = 0.f; q1[r][v] = 0.f; }
        }
    }

    // occurrence counts of the batch's entities for the row backward (fire-and-forget atomics of the first position slice)
    if (A.occ && sl == 0) {
        if (tid < TI && i0 + tid < A.B) {
            atomicAdd(A.occ + A.occ_sample[3 * (int64_t)(i0 + tid)], 1);
            atomicAdd(A.occ + A.occ_sample[3 * (int64_t)(i0 + tid) + 2], 1);
        }
        for (int p = bx * WG + tid; p < A.P; p += n_bx * WG) atomicAdd(A.occ + A.pool[p], 1);
    }

    // positions used by at least one row of the tile, compacted into LDS
    MKB_TRACE_T(tr_t0);
    int n_act = 0;
    for (int base = 0; base < Pn; base += WG) {
        const int p = A.p_lo + sl + (base + tid) * nsl;
        unsigned m_own = 0;
        int ent_p = 0;
        if (base + tid < Pn) {
            // the eight multiplicities and the position's pool id are requested together, unconditionally (rows past the batch
            // read row B - 1 and are masked): as conditional loads each sat in a basic block of its own and was waited for there --
            // nine round trips per list pass (round 5, ISA), ten with the id fetched behind the compaction
            unsigned c[TI];
#pragma unroll
            for (int r = 0; r < TI; ++r) c[r] = A.cnt[(int64_t)min(i0 + r, A.B - 1) * A.P + p];
            ent_p = (int)A.pool[p];
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                if (i0 + r >= A.B) c[r] = 0;
                m_own |= (c[r] != 0) ? (1u << r) : 0u;
                // entries no row uses are defined as 0 (the pair loop below never visits them): no memset launch
                if (c[r] == 0 && i0 + r < A.B) A.S[(int64_t)(i0 + r) * A.P + p] = 0.f;
            }
        }
        int tot;
        const int slot = n_act + wg_compact_slot<NW>(m_own != 0, s_wave_cnt, &tot);
        if (m_own != 0) {
            s_pos[slot] = p;
            s_mask[slot] = m_own;
            s_row[slot] = ent_p;
        }
        n_act += tot;
        __syncthreads();
    }

    MKB_TRACE_T(tr_t1);
    const int n_mine = n_act;
    const int j_last = n_mine - 1;

    // Candidate rows stream through a kRing-deep register ring; loads are UNCONDITIONAL (index clamped to the last
    // position) so that the compiler can count outstanding loads instead of draining them.
    float xr0[kRing][KPT], xr1[kRing][KPT];
    auto load_x = [&](int j, float (&d0)[KPT], float (&d1)[KPT]) {
        const float *x = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(s_row[j]) * A.De;
        load_units<CP, KPT>(x, A.d, NU, u0, d0, d1);
    };
    if (n_mine > 0) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) load_x(min(s, j_last), xr0[s], xr1[s]);
    }
    for (int jb = 0; jb < n_mine; jb += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int j = jb + s;
            const int jj = j % kSlab, buf = (j / kSlab) & 1;
            const unsigned m = (j < n_mine) ? __builtin_amdgcn_readfirstlane(s_mask[min(j, j_last)]) : 0u;
            float x0[KPT], x1[KPT];
#pragma unroll
            for (int v = 0; v < KPT; ++v) { x0[v] = xr0[s][v]; x1[v] = xr1[s][v]; }
            load_x(min(j + kRing, j_last), xr0[s], xr1[s]);
            float part[TI];
            // Two bodies: the branch-free one evaluates all 8 rows as one basic block (independent chains interleave,
            // no per-row scalar branch); rows that do not use the position are computed and never stored.  Positions
            // only a few rows use keep the per-row branches.
            auto rows = [&](auto dense_c) {
                constexpr bool DENSE = decltype(dense_c)::value;
#pragma unroll
                for (int r = 0; r < TI; ++r) {
                    part[r] = 0.f;
                    if (DENSE || (m & (1u << r))) {  // out-of-range units hold q = x = 0 and contribute exactly 0
                        if constexpr (CP && KPT % 2 == 0) {
                            // (start from the first term: 0 + t is a real v_pk_add, -0 semantics forbid folding it)
                            f2 acc = pair_term_cmod2(f2{q0[r][0], q0[r][1]}, f2{q1[r][0], q1[r][1]}, f2{x0[0], x0[1]},
                                                     f2{x1[0], x1[1]});
#pragma unroll
                            for (int v = 2; v < KPT; v += 2)
                                acc += pair_term_cmod2(f2{q0[r][v], q0[r][v + 1]}, f2{q1[r][v], q1[r][v + 1]},
                                                       f2{x0[v], x0[v + 1]}, f2{x1[v], x1[v + 1]});
                            part[r] = acc.x + acc.y;
                        } else {
#pragma unroll
                            for (int v = 0; v < KPT; ++v) {
                                if constexpr (CP) part[r] += pair_term_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]});
                                else part[r] += pair_term_real<MODEL, HEAD>(q0[r][v], x0[v], A.kd);
                            }
                        }
                    }
                }
            };
            if (__builtin_popcount(m) >= kDense) rows(std::true_type{});
            else rows(std::false_type{});
            float t0, t1;
            reduce8_wave(part, t0, t1);
            if ((lane & 15) == 0) {
                const int R = lane >> 4, r = 4 * (R >> 1) + 2 * (R & 1);
                s_part[buf][jj][wave][r] = t0;
                s_part[buf][jj][wave][r + 1] = t1;
            }
            if (j < n_mine && (jj == kSlab - 1 || j == j_last)) {  // wave-uniform: combine <= kSlab positions
                __syncthreads();  // double-buffered s_part: one barrier per batch
                const int j0 = j - jj, nb = jj + 1;
                for (int e = tid; e < nb * TI; e += WG) {  // (a 1-wave workgroup has fewer lanes than entries)
                    const int cj = e / TI, r = e % TI;
                    const int a = j0 + cj;
                    if (s_mask[a] & (1u << r)) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) sum += s_part[buf][cj][w][r];
                        if constexpr (MODEL == MKB_PROTATE) sum *= A.modulus[0];  // gamma - modulus * sum (protate.py:91)
                        A.S[(int64_t)(i0 + r) * A.P + s_pos[a]] = A.c0 + A.c1 * sum;
                    }
                }
            }
        }
    }
    MKB_TRACE_OUT(A, 0, tr_t0, tr_t1, wall_clock64(), n_mine);
}

template <int MODEL, bool HEAD, int KPT, int NW>
__global__ __launch_bounds__(NW * 64) void pool_fwd_kernel(PoolArgs A) {
    extern __shared__ __attribute__((aligned(16))) int lds_fwd[];  // sized by the launch: 3 * ceil(P / nslices) words
    pool_fwd_body<MODEL, HEAD, KPT, NW>(A, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y, lds_fwd);
}

// ------------------------------------------------------------------------------------------------ backward: dq
// Body of one (row tile, position slice) workgroup of the merged backward kernel (pool_bwd_kernel below).
template <int MODEL, bool HEAD, int KPT, int NW>
__device__ __forceinline__ void pool_bwd_q_body(const PoolArgs &A, const int block, int *lds_dyn) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int WG = NW * 64;
    // ALL LDS is dynamic (10 * P + 32 words): a static __shared__ in front of the dynamic region can shift its base
    // off 16 bytes, and the ds_read_b128 of s_g would then be replayed (cdna guide, G17)
    const int row_tiles = (A.B + TI - 1) / TI;
    const int nsl = A.q_slices, sl = block / row_tiles;  // position slice: p = slice + nslices * i, as in the forward kernel
    const int Pn = (A.P - sl + nsl - 1) / nsl;
    const int Pcap = (A.P + nsl - 1) / nsl;
    float (*s_g)[TI] = reinterpret_cast<float (*)[TI]>(lds_dyn);   // [Pcap][8] gradient seeds per active position
    int *s_row = lds_dyn + TI * Pcap;
    unsigned *s_mask = reinterpret_cast<unsigned *>(lds_dyn + (TI + 1) * Pcap);
    int *s_wave_cnt = lds_dyn + (TI + 2) * Pcap;
    float *s_red = reinterpret_cast<float *>(lds_dyn + (TI + 2) * Pcap + 16);

    const int tid = threadIdx.x;
    const int i0 = (block - sl * row_tiles) * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;

    // the tile's query rows first: their latency overlaps the list build
    float q0[TI][KPT], q1[TI][KPT], dq0[TI][KPT], dq1[TI][KPT];
#pragma unroll
    for (int r = 0; r < TI; ++r) {
        load_units<CP, KPT>(A.Q + (int64_t)min(i0 + r, A.B - 1) * A.De, A.d, NU, u0, q0[r], q1[r]);
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if (i0 + r >= A.B) { q0[r][v] = 0.f; q1[r][v] = 0.f; }
            dq0[r][v] = 0.f;
            dq1[r][v] = 0.f;
        }
    }
    MKB_TRACE_T(tr_t0);
    int n_act = 0;
    for (int base = 0; base < Pn; base += WG) {
        const int p = sl + (base + tid) * nsl;
        unsigned m_own = 0;
        float g_own[TI];
#pragma unroll
        for (int r = 0; r < TI; ++r) g_own[r] = 0.f;
        if (base + tid < Pn) {
            unsigned c[TI];
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                const bool in = i0 + r < A.B;
                c[r] = in ? A.cnt[(int64_t)(i0 + r) * A.P + p] : 0;
                g_own[r] = in ? A.G[(int64_t)(i0 + r) * A.P + p] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                m_own |= (c[r] != 0) ? (1u << r) : 0u;
                g_own[r] = (c[r] != 0) ? g_own[r] : 0.f;  // the branch-free body relies on g = 0 for unused pairs
            }
        }
        int tot;
        const int slot = n_act + wg_compact_slot<NW>(m_own != 0, s_wave_cnt, &tot);
        if (m_own != 0) {
            s_mask[slot] = m_own;
            s_row[slot] = (int)A.pool[p];
#pragma unroll
            for (int r = 0; r < TI; ++r) s_g[slot][r] = g_own[r];
        }
        n_act += tot;
        __syncthreads();
    }

    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;
    float extra = 0.f;

    MKB_TRACE_T(tr_t1);
    const int n_mine = n_act;
    const int j_last = n_mine - 1;
    float xr0[kRing][KPT], xr1[kRing][KPT];
    auto load_x = [&](int j, float (&d0)[KPT], float (&d1)[KPT]) {
        const float *x = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(s_row[j]) * A.De;
        load_units<CP, KPT>(x, A.d, NU, u0, d0, d1);
    };
    if (n_mine > 0) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) load_x(min(s, j_last), xr0[s], xr1[s]);
    }
    for (int jb = 0; jb < n_mine; jb += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int j = jb + s;
            const int a = min(j, j_last);
            const unsigned m = (j < n_mine) ? __builtin_amdgcn_readfirstlane(s_mask[a]) : 0u;
            float g[TI];
#pragma unroll
            for (int r = 0; r < TI; ++r) g[r] = uniform_f32(s_g[a][r]);  // two ds_read_b128, same address in every lane -> SGPRs
            float x0[KPT], x1[KPT];
#pragma unroll
            for (int v = 0; v < KPT; ++v) { x0[v] = xr0[s][v]; x1[v] = xr1[s][v]; }
            load_x(min(j + kRing, j_last), xr0[s], xr1[s]);
            // branch-free body for positions most rows use (see the forward kernel): unused pairs carry g = 0 and add 0
            auto rows = [&](auto dense_c) {
                constexpr bool DENSE = decltype(dense_c)::value;
#pragma unroll
                for (int r = 0; r < TI; ++r) {
                    if (DENSE || (m & (1u << r))) {
                        if constexpr (CP && KPT % 2 == 0) {
#pragma unroll
                            for (int v = 0; v < KPT; v += 2) {
                                f2 ar = f2{dq0[r][v], dq0[r][v + 1]}, ai = f2{dq1[r][v], dq1[r][v + 1]};
                                pair_bwd_cmod2(f2{q0[r][v], q0[r][v + 1]}, f2{q1[r][v], q1[r][v + 1]}, f2{x0[v], x0[v + 1]},
                                               f2{x1[v], x1[v + 1]}, g[r], ar, ai);
                                dq0[r][v] = ar.x; dq0[r][v + 1] = ar.y;
                                dq1[r][v] = ai.x; dq1[r][v + 1] = ai.y;
                            }
                        } else
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
            };
            if (__builtin_popcount(m) >= kDense) rows(std::true_type{});
            else rows(std::false_type{});
        }
    }
    MKB_TRACE_T(tr_t2);
    float *dQs = A.dQ + (int64_t)sl * A.B * A.De;
#pragma unroll
    for (int r = 0; r < TI; ++r)
        if (i0 + r < A.B) store_units<CP, KPT>(dQs + (int64_t)(i0 + r) * A.De, A.d, NU, u0, dq0[r], dq1[r]);
    if constexpr (MODEL == MKB_PROTATE) {  // d score / d modulus = - sum_k |sin z|   (protate.py:91)
        extra = wave_sum(extra);
        if ((tid & 63) == 0) s_red[tid >> 6] = extra;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < NW; ++w) s += s_red[w];
            atomicAdd(A.g_modulus, -s);
        }
    }
    MKB_TRACE_OUT(A, 1, tr_t0, tr_t1, tr_t2, n_mine);
}

// ------------------------------------------------------------------------------------------------ backward: dx
template <int MODEL, bool HEAD, int KPT, int NW>
__device__ __forceinline__ void pool_bwd_x_body(const PoolArgs &A, const int block, int *lds_dyn) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int WG = NW * 64;
    // all LDS dynamic: 10 * rows_per + 16 words

    const int tid = threadIdx.x;
    // 1-D grid, tile-major: the workgroups of the low position tiles (used by every row: the heavy ones) are
    // dispatched first; the light / empty tiles fill in behind them.
    // ... except the fringe: the first tiles past position P / 2 (every row takes its first P / 2 surviving candidates,
    // so those tiles are used by the rows the filter hit, a few positions each).  Their workgroups are latency-bound
    // (little math per streamed query row) and starve when they share a CU with an older heavy workgroup -- the CU
    // issues oldest-first -- so they are dispatched FIRST: measured tail 103 -> 78 us at the headline shape.
    const int nsl = A.x_slices, sl = block % nsl;
    const int n_tiles = (A.P + TI - 1) / TI, heavy_tiles = min(n_tiles, (A.P / 2 + TI - 1) / TI);
    const int fringe = min(2, n_tiles - heavy_tiles);
    const int bt = block / nsl;
    const int tile = bt < fringe ? heavy_tiles + bt : (bt < heavy_tiles + fringe ? bt - fringe : bt);
    const int p0 = tile * TI;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = tid * KPT;
    const int rows_per = (A.B + nsl - 1) / nsl;
    const int r_lo = sl * rows_per, r_hi = min(A.B, r_lo + rows_per);
    float (*s_g)[TI] = reinterpret_cast<float (*)[TI]>(lds_dyn);          // [rows_per][8]
    int *s_i = lds_dyn + TI * rows_per;                                   // batch rows of the slice that use the tile
    unsigned *s_mask = reinterpret_cast<unsigned *>(lds_dyn + (TI + 1) * rows_per);  // bit t: row uses position p0 + t
    int *s_wave_cnt = lds_dyn + (TI + 2) * rows_per;
    // the tile's candidate rows first: their latency overlaps the row-list build
    float x0[TI][KPT], x1[TI][KPT], dx0[TI][KPT], dx1[TI][KPT];
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const bool pin = p0 + t < A.P;
        load_units<CP, KPT>(A.ent + (pin ? A.pool[p0 + t] : 0) * A.De, A.d, NU, u0, x0[t], x1[t]);
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if (!pin) { x0[t][v] = 0.f; x1[t][v] = 0.f; }
            dx0[t][v] = 0.f;
            dx1[t][v] = 0.f;
        }
    }
    MKB_TRACE_T(tr_t0);

    int n_rows = 0;
    for (int base = r_lo; base < r_hi; base += WG) {
        const int i_own = base + tid;
        unsigned m_own = 0;
        float g_own[TI];
#pragma unroll
        for (int t = 0; t < TI; ++t) g_own[t] = 0.f;
        if (i_own < r_hi) {
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                if (p0 + t < A.P) {
                    const unsigned c = A.cnt[(int64_t)i_own * A.P + p0 + t];
                    g_own[t] = (c != 0) ? A.G[(int64_t)i_own * A.P + p0 + t] : 0.f;  // g = 0: the branch-free body adds 0
                    m_own |= (c != 0) ? (1u << t) : 0u;
                }
            }
        }
        int tot;
        const int slot = n_rows + wg_compact_slot<NW>(m_own != 0, s_wave_cnt, &tot);
        if (m_own != 0) {
            s_i[slot] = i_own;
            s_mask[slot] = m_own;
#pragma unroll
            for (int t = 0; t < TI; ++t) s_g[slot][t] = g_own[t];
        }
        n_rows += tot;
        __syncthreads();
    }

    MKB_TRACE_T(tr_t1);
    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;

    float qr0[kRing][KPT], qr1[kRing][KPT];
    auto load_q = [&](int j, float (&d0)[KPT], float (&d1)[KPT]) {
        const float *q = A.Q + (int64_t)__builtin_amdgcn_readfirstlane(s_i[j]) * A.De;
        load_units<CP, KPT>(q, A.d, NU, u0, d0, d1);
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
            const int jc = min(j, j_last);
            const unsigned m = (j < n_rows) ? __builtin_amdgcn_readfirstlane(s_mask[jc]) : 0u;
            float g[TI];
#pragma unroll
            for (int t = 0; t < TI; ++t) g[t] = uniform_f32(s_g[jc][t]);  // wave-uniform: keep the seeds in SGPRs
            float q0[KPT], q1[KPT];
#pragma unroll
            for (int v = 0; v < KPT; ++v) { q0[v] = qr0[s][v]; q1[v] = qr1[s][v]; }
            load_q(min(j + kRing, j_last), qr0[s], qr1[s]);
            // branch-free body for rows that use most of the tile (see the forward kernel): unused pairs carry g = 0
            auto positions = [&](auto dense_c) {
                constexpr bool DENSE = decltype(dense_c)::value;
#pragma unroll
                for (int t = 0; t < TI; ++t) {
                    if (DENSE || (m & (1u << t))) {
                        if constexpr (CP && KPT % 2 == 0) {  // accumulates -dx (the dq sign); negated once at the store
#pragma unroll
                            for (int v = 0; v < KPT; v += 2) {
                                f2 ar = f2{dx0[t][v], dx0[t][v + 1]}, ai = f2{dx1[t][v], dx1[t][v + 1]};
                                pair_bwd_cmod2(f2{q0[v], q0[v + 1]}, f2{q1[v], q1[v + 1]}, f2{x0[t][v], x0[t][v + 1]},
                                               f2{x1[t][v], x1[t][v + 1]}, g[t], ar, ai);
                                dx0[t][v] = ar.x; dx0[t][v + 1] = ar.y;
                                dx1[t][v] = ai.x; dx1[t][v + 1] = ai.y;
                            }
                        } else
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
            };
            if (__builtin_popcount(m) >= kDense) positions(std::true_type{});
            else positions(std::false_type{});
        }
    }
    MKB_TRACE_T(tr_t2);
    // dx of this (position tile, row slice) goes straight into the table gradient: x_slices (8) fp32 atomics per
    // element of a used pool row (~5 M per step, spread over the kernel), instead of a [slices, P, De] partial buffer
    // plus a reduction kernel.  Tiles nobody uses (n_rows == 0) write nothing.
    if (n_rows > 0) {
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            if (p0 + t < A.P && u0 < NU) {
                float *row = A.g_ent + A.pool[p0 + t] * A.De;
                const float sgn = (CP && KPT % 2 == 0) ? -1.f : 1.f;  // the packed path accumulated -dx
#pragma unroll
                for (int v = 0; v < KPT; ++v) {
                    atomicAdd(row + u0 + v, sgn * dx0[t][v]);
                    if constexpr (CP) atomicAdd(row + A.d + u0 + v, sgn * dx1[t][v]);
                }
            }
        }
    }
    MKB_TRACE_OUT(A, 2, tr_t0, tr_t1, tr_t2, n_rows);
}

// ------------------------------------------------------------------------------------------------ backward: one launch
// The dq pass and the dx pass read the same inputs and write disjoint outputs, so they are one grid.  One launch instead
// of two saves a kernel boundary (~15 us between two large kernels on this part) and keeps two workgroups resident on
// every CU for the whole launch.  Dispatch order (a CU issues oldest-first, so the order is the schedule): the first
// q_first dq workgroups (on most CUs), then the dx pass (fringe, heavy, light tiles: the heavy ones take the CUs' second
// slots), then the remaining dq workgroups, which replace the first ones as they retire.  x_blocks == 0 or no dq blocks
// runs one pass alone (A/B measurements).
template <int MODEL, bool HEAD, int KPT, int NW>
__global__ __launch_bounds__(NW * 64) void pool_bwd_kernel(PoolArgs A) {
    extern __shared__ __attribute__((aligned(16))) int lds_bwd[];
    const int b = (int)blockIdx.x;
    if (b < A.q_first) pool_bwd_q_body<MODEL, HEAD, KPT, NW>(A, b, lds_bwd);
    else if (b < A.q_first + A.x_blocks) pool_bwd_x_body<MODEL, HEAD, KPT, NW>(A, b - A.q_first, lds_bwd);
    else pool_bwd_q_body<MODEL, HEAD, KPT, NW>(A, b - A.x_blocks, lds_bwd);
}

// ------------------------------------------------------------------------------------------------ backward: small problems
// pool_bwd_wave: when the whole backward is a handful of row groups (Umls: 256 rows x 32 positions x 64 dims = 0.5 MFLOP), the
// single-pass kernel's skeleton IS the launch: two 16-wave workgroups walking a 16-phase ring took 43 us there, the two-pass
// kernel 36.  Here one WAVE = (tile of 8 rows, slice of <= 64 pool positions, 64 * KPT units): no LDS, no ring, no workgroup.
// Lane j holds the 8 seeds and the table offset of the slice's j-th position (plain [B, P] seed layout); the used positions
// (any seed != 0) are walked off a ballot mask with their candidate rows requested four ahead.  Every pair term is evaluated
// once: dq accumulates in registers and is stored to the slice's dQ partial buffer (the row backward adds the slices up),
// the position's dx goes to the table gradient with one fp32 atomic per element (row tiles x used positions x units of
// them: 65 k at the Umls shape).
template <int MODEL, bool HEAD, int KPT>
__global__ __launch_bounds__(64) void pool_bwd_wave_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int R = 4;  // candidate rows in flight
    const int lane = threadIdx.x;
    const int tile = (int)blockIdx.x, sl = (int)blockIdx.y, chunk = (int)blockIdx.z;
    const int nsl = A.q_slices;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = (chunk * 64 + lane) * KPT;
    const int i0 = tile * TI;
    const int npos = (A.P - sl + nsl - 1) / nsl;  // positions p = sl + nsl * j, j < npos <= 64 (host: nsl >= ceil(P / 64))
    float q0[TI][KPT], q1[TI][KPT], dq0[TI][KPT], dq1[TI][KPT];
#pragma unroll
    for (int r = 0; r < TI; ++r) load_units<CP, KPT>(A.Q + (int64_t)min(i0 + r, A.B - 1) * A.De, A.d, NU, u0, q0[r], q1[r]);
    const int p_own = sl + nsl * lane;
    const bool valid = lane < npos && p_own < A.P;
    float gv[TI];
    unsigned nz = 0;
#pragma unroll
    for (int r = 0; r < TI; ++r) {
        gv[r] = (valid && i0 + r < A.B) ? A.G[(int64_t)(i0 + r) * A.P + p_own] : 0.f;
        nz |= __float_as_uint(gv[r]) << 1;  // (+-0 = the row does not use the position: the loss kernel writes 0 there)
    }
    const int64_t off = (valid ? A.pool[p_own] : 0) * A.De;
    const int off_lo = (int)(unsigned)off, off_hi = (int)(off >> 32);
    unsigned long long req = __ballot(nz != 0);  // positions not requested yet (wave-uniform)
#pragma unroll
    for (int r = 0; r < TI; ++r) {
#pragma unroll
        for (int v = 0; v < KPT; ++v) {
            if (i0 + r >= A.B) { q0[r][v] = 0.f; q1[r][v] = 0.f; }
            dq0[r][v] = 0.f; dq1[r][v] = 0.f;
        }
    }
    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;
    float extra = 0.f;
    auto take = [&]() {  // the next used position of the slice, or -1
        if (!req) return -1;
        const int j = (int)__builtin_ctzll(req);
        req &= req - 1ull;
        return j;
    };
    auto offset_of = [&](int j) {
        return ((int64_t)__builtin_amdgcn_readlane(off_hi, j) << 32) | (int64_t)(unsigned)__builtin_amdgcn_readlane(off_lo, j);
    };
    float xr0[R][KPT], xr1[R][KPT];
    int jr[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        jr[k] = take();
        load_units<CP, KPT>(A.ent + offset_of(max(jr[k], 0)), A.d, NU, u0, xr0[k], xr1[k]);
    }
    for (bool more = true; more;) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int j = jr[k];
            if (j < 0) { more = false; break; }
            float x0[KPT], x1[KPT];
#pragma unroll
            for (int v = 0; v < KPT; ++v) { x0[v] = xr0[k][v]; x1[v] = xr1[k][v]; }
            const int64_t xoff = offset_of(j);
            jr[k] = take();
            load_units<CP, KPT>(A.ent + offset_of(max(jr[k], 0)), A.d, NU, u0, xr0[k], xr1[k]);
            float dx0[KPT], dx1[KPT];
#pragma unroll
            for (int v = 0; v < KPT; ++v) { dx0[v] = 0.f; dx1[v] = 0.f; }
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                const float g = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(gv[r]), j));
                if ((__float_as_uint(g) << 1) == 0u) continue;  // (wave-uniform)
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
            if (u0 < NU) {
                float *grow = A.g_ent + xoff;
#pragma unroll
                for (int v = 0; v < KPT; ++v) {
                    atomicAdd(grow + u0 + v, dx0[v]);
                    if constexpr (CP) atomicAdd(grow + A.d + u0 + v, dx1[v]);
                }
            }
        }
    }
    float *dQs = A.dQ + (int64_t)sl * A.B * A.De;
#pragma unroll
    for (int r = 0; r < TI; ++r)
        if (i0 + r < A.B) store_units<CP, KPT>(dQs + (int64_t)(i0 + r) * A.De, A.d, NU, u0, dq0[r], dq1[r]);
    if constexpr (MODEL == MKB_PROTATE) {  // d score / d modulus = - sum_k |sin z|   (protate.py:91)
        extra = wave_sum(extra);
        if (lane == 0 && extra != 0.f) atomicAdd(A.g_modulus, -extra);
    }
}

// ------------------------------------------------------------------------------------------------ backward: single pass
// pool_bwd1: every (row, pool position, unit) pair term is evaluated ONCE and feeds both gradients.
//   workgroup = (dim slice of 64*KPT units, group of 16*tiles_per_wave row tiles, block of <= 64*halves pool positions;
//                slot (h, l) of the block is position p = block + nblocks * (l * halves + h): blocks and halves are
//                interleaved so that each gets the same share of the dense prefix p < K);
//   wave      = one row tile at a time: q[8], dq[8] of its slice in registers.  Lane l holds the 8 gradient seeds and
//               the entity id of slot (h, l) in VGPRs; per position they reach SGPRs with v_readlane (no LDS slab, no
//               scalar-memory latency in the loop).  The 8 pair terms of a position update dq[r] and one running dx;
//   dx        = accumulated in the workgroup's LDS array s_dx[slot][lane][component].  LDS float atomics are far too
//               slow for this (ds_add_f32 ~ 128 cycles per wave instruction: measured 396 us for the launch), so the
//               16 waves walk the block in 16 lock-step PHASES: in phase t wave w owns chunk (w + t) mod 16 (every
//               16/halves-th lane of one half: equal shares of the dense prefix) and updates it with plain
//               ds_read_b128 / ds_write_b128; one barrier per phase hands the chunks on;
//   flush     = after the last phase the LDS array goes to the table gradient rows (one fp32 atomic per element and row
//               group: 8 per element at the headline shape, as before); dq partial stored once per wave and block.
// Nothing is recomputed: RotatE 9 packed ops + 2 v_rsq per two complex dims (the two-pass kernels: 14 + 4).
#ifndef MKB_HANDON_PRIO
#define MKB_HANDON_PRIO 0
#endif
#ifndef MKB_BWD1_Q_EARLY
#define MKB_BWD1_Q_EARLY 1
#endif
constexpr int kBwd1Waves = 16;
constexpr int kChunkStride = 1;                         // chunks (= phases) between a wave and the next wave of the chain
constexpr int kChunks = kBwd1Waves * kChunkStride;      // chunks per block = phases per tile

template <int NC> struct AccVec;
template <> struct AccVec<1> { typedef float type; };
template <> struct AccVec<2> { typedef float2 type; };
template <> struct AccVec<4> { typedef float4 type; };

template <int MODEL, bool HEAD, int KPT, bool DENSE>  // DENSE: dense pass + chain-free fringe (A.dense_lanes > 0); else the general pass
__global__ __launch_bounds__(kBwd1Waves * 64) void pool_bwd1_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int NC = KPT * (CP ? 2 : 1);  // floats per lane and position
    constexpr int NW = kBwd1Waves, WG = NW * 64;
    typedef typename AccVec<NC>::type acc_t;
    extern __shared__ __attribute__((aligned(16))) int lds1[];
    const int halves = A.pb_halves, cap = halves * 64;  // halves in {1, 2, 4, 8}
    // accumulator slots: all of the block's (general pass), or the dense lanes of every half only -- the fringe's dx never
    // passes through LDS then, and the space holds the waves' candidate-row rings instead
    const int acc_slots = DENSE ? halves * A.dense_lanes : cap;
    // DENSE: [128-byte header][accumulator][row images]; behind the dense pass the same space holds the workgroup's 128 query
    // rows for the fringe's dx half (launch_bwd1 sizes the allocation for the larger of the two uses)
    constexpr int HDR = DENSE ? 32 : 0;  // (ints)
    acc_t *s_dx = reinterpret_cast<acc_t *>(lds1 + HDR);                                                            // [acc_slots][64]
    unsigned long long *s_used = reinterpret_cast<unsigned long long *>(DENSE ? lds1 : lds1 + (size_t)acc_slots * NC * 64);  // [halves <= 8]
    int *s_done = (DENSE ? lds1 : lds1 + (size_t)acc_slots * NC * 64) + 16;  // [NW] phases finished by each wave (hand-off chain, see below)
    acc_t *s_x = s_dx + (size_t)acc_slots * 64;  // DENSE: [acc_slots][64] the workgroup's slices of the dense positions' candidate rows
    acc_t *s_q = s_dx;                           // DENSE, one tile per wave: [16 waves x 8 rows][64] query rows, once the dense pass is over
    // DENSE: pool id of every slot of the block, behind everything else (launch_bwd1: + cap * 8 bytes): the fringe's dx half
    // asks for a slot's candidate row without a round trip for its id first
    int64_t *s_ids = reinterpret_cast<int64_t *>(lds1 + HDR + A.lds_ids_off);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // tell the compiler it is wave-uniform (scalar control flow)
    int b = (int)blockIdx.x;
    const int s = b % A.dim_slices; b /= A.dim_slices;
    const int npb = A.q_slices, pb = b % npb, rg = b / npb;
    const int NU = CP ? A.d : (int)A.De;
    const int u0 = (s * 64 + lane) * KPT;
    // The block's slots are cut into kChunks = 16 chunks, one per wave and phase: chunk c = lanes l == c % cph (mod cph) of half c / cph
    const int cph = kChunks / halves;
    MKB_TRACE_T(tr_t0);
    MKB_TRACE_ONLY(unsigned long long tr_hand = 0, tr_setup = 0, tr_items = 0, tr_dense = 0, tr_pro = 0, tr_p2 = 0, tr_p2rows = 0, tr_p2slots = 0, tr_p2batch = 0; const unsigned long long tr_c0 = __builtin_readcyclecounter();)

    // The wave's first row tile: its 8 query-row slices are requested at ENTRY, ahead of the pool ids and the candidate-row images
    // below (two dependent round trips) and the barrier behind them -- round 6: they used to be requested behind that barrier, a
    // third round trip in series in a workgroup that owns its CU alone (nothing else hides it).  MKB_BWD1_Q_EARLY=0: A/B builds.
    const int row_tiles = (A.B + TI - 1) / TI;
    float q0[TI][KPT], q1[TI][KPT];
    if constexpr (MKB_BWD1_Q_EARLY != 0) {
        const int i00 = ((rg * A.tiles_per_wave) * NW + wave) * TI;
        if (i00 < A.B) {
#pragma unroll
            for (int r = 0; r < TI; ++r) load_units_raw<CP, KPT>(A.Q + (int64_t)min(i00 + r, A.B - 1) * A.De, A.d, NU, u0, q0[r], q1[r]);
        }
    }
    for (int e = tid * 4; e < acc_slots * NC * 64; e += WG * 4)
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(s_dx) + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < halves) s_used[tid] = 0ull;
    if (tid < NW) s_done[tid] = 0;
    if constexpr (DENSE) {
        for (int sl = tid; sl < cap; sl += WG) {
            const int p = pb + npb * ((sl & 63) * halves + (sl >> 6));
            s_ids[sl] = p < A.P ? A.pool[p] : 0;
        }
        // the candidate rows of the dense positions, one image per accumulator slot: wave w fills the slots of chunk w
        // (the chunk it owns in phase 0); the loads of the chunk are issued together
        const int cph0 = kChunks / halves, lc0 = __builtin_ctz((unsigned)cph0), nd0 = A.dense_lanes >> lc0;
        const int h0 = wave >> lc0, l0 = wave & (cph0 - 1);
        for (int k = 0; k < nd0; k += 4) {
            // (round 6, from the ISA: written as "four load_units, then four stores" this was compiled into four SERIAL
            // (scalar id load -> wait -> row loads -> wait -> LDS store) sequences -- the select inside load_units wants the data at
            // once and the scheduler does not move loads across a wait -- eight dependent round trips in a prologue nothing hides.
            // The four ids first, then the four rows' raw loads, a scheduling fence, the masking at the store.)
            float xa[4][KPT], xb[4][KPT];
            int64_t idc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j0 = l0 + (min(k + c, nd0 - 1) << lc0);
                idc[c] = A.pool[pb + A.q_slices * (j0 * halves + h0)];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                load_units_raw<CP, KPT>(A.ent + idc[c] * A.De, A.d, NU, u0, xa[c], xb[c]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (k + c >= nd0) break;
                const int j0 = l0 + ((k + c) << lc0);
                if (u0 >= NU) {  // (lanes past the row's end hold unit 0's data: the images carry zeros there)
#pragma unroll
                    for (int v = 0; v < KPT; ++v) { xa[c][v] = 0.f; xb[c][v] = 0.f; }
                }
                acc_t v;
                if constexpr (NC == 1) v = xa[c][0];
                else if constexpr (NC == 2 && !CP) { v.x = xa[c][0]; v.y = xa[c][1]; }
                else if constexpr (NC == 2) { v.x = xa[c][0]; v.y = xb[c][0]; }
                else if constexpr (NC == 4 && !CP) { v.x = xa[c][0]; v.y = xa[c][1]; v.z = xa[c][KPT - 2]; v.w = xa[c][KPT - 1]; }
                else { v.x = xa[c][0]; v.y = xa[c][1]; v.z = xb[c][0]; v.w = xb[c][1]; }
                s_x[(size_t)(h0 * A.dense_lanes + j0) * 64 + lane] = v;
            }
        }
    }
    __syncthreads();
    MKB_TRACE_T(tr_t1);

    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;
    for (int t = 0; t < A.tiles_per_wave; ++t) {
        const int tile = (rg * A.tiles_per_wave + t) * NW + wave;
        const bool have = tile < row_tiles;  // (wave-uniform; a wave without a tile only keeps the barriers)
        const int i0 = tile * TI;
        float dq0[TI][KPT], dq1[TI][KPT];
        if (MKB_BWD1_Q_EARLY == 0 || t > 0) {  // (tile 0's rows were requested at entry)
#pragma unroll
            for (int r = 0; r < TI; ++r) {
                // (all eight rows' loads first, the selects behind them: a select next to its load makes the wave wait there, and the
                // eight round trips ran one after the other)
                if (have) load_units_raw<CP, KPT>(A.Q + (int64_t)min(i0 + r, A.B - 1) * A.De, A.d, NU, u0, q0[r], q1[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < TI; ++r) {
#pragma unroll
            for (int v = 0; v < KPT; ++v) {
                if (!have || i0 + r >= A.B || u0 >= NU) { q0[r][v] = 0.f; q1[r][v] = 0.f; }
                dq0[r][v] = 0.f;
                dq1[r][v] = 0.f;
            }
        }
        float extra = 0.f;
        // lanes of chunk c of a half: l0, l0 + cph, ... (bit pattern with every cph-th bit set, shifted by l0)
        // (cph = 16 / halves in {16, 8, 4, 2}; all-ones / (2^cph - 1) = 1 at every cph-th bit.  4 and 2 -- four or eight halves,
        // i.e. more than 128 positions per block -- were missing from an enumerated table here until round 3)
        const unsigned long long cm = ~0ull / ((1ull << cph) - 1ull);
        // Hand-off between phases.  Wave v owns chunk (4 v + g) mod 64 in (global) phase g; that chunk was wave v + 1's in
        // phase g - 4, wave v + 2's in phase g - 8, ...: wave v may enter phase g as soon as wave v + 1 has finished phase
        // g - 4 (which, by the same rule, implies every earlier owner has).  Four chunks per wave instead of one give the chain
        // 3 phases of slack per link (12-16 between the four waves that share a SIMD): in steady state nobody waits and the
        // SIMD always has several runnable waves (with one chunk per wave, every wave spent half its loop time waiting for
        // its predecessor -- per-wave cycle accounts of tools/wgtrace.py -- and the waves of a SIMD ran one at a time).  So instead of a workgroup barrier per phase -- 16 drains per
        // tile with the SIMDs running out of ready waves at each (PMC: waves parked 55 % of their lifetime, VALU busy 44 %)
        // -- each wave publishes its finished-phase count in LDS and waits for its ONE predecessor only; the waves fall
        // into a staggered pipeline.  LDS operations of a wave are performed in order, so the count lands after the data.
        // A tile takes one or two passes over the 16 chunks: the DENSE pass (lanes [0, dense_lanes) of every half: positions
        // of the pool's dense prefix, walked without masks or stream order) and the general pass (the other lanes: the sparse
        // fringe).  The chunk rotation simply continues from one pass into the next.
        constexpr int n_pass = DENSE ? 2 : 1;
        const int Ld = (DENSE && have) ? A.dense_lanes : 0;  // (wave-uniform)
        const int gd0 = t * n_pass * kChunks, g0 = gd0 + (n_pass - 1) * kChunks, pred = (wave + 1) & (NW - 1);
        auto hand_on = [&](int finished) {
            // (A wave without a row tile walks the chain like the others.  It used to publish "infinitely far ahead", which cuts
            // the ring open: its successor in the chain is then bounded from below only and may run 15 phases ahead of ITS
            // successor -- the phase in which the two own the same chunk.)
            if (lane == 0) __hip_atomic_store(&s_done[wave], finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            MKB_TRACE_ONLY(const unsigned long long th0 = __builtin_readcyclecounter();)
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_done[pred], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < finished - (kChunkStride - 1))
                __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            MKB_TRACE_ONLY(tr_hand += __builtin_readcyclecounter() - th0;)
            // Fair share of the SIMD: the hardware issues oldest-first, which lets the oldest wave of each SIMD run a phase
            // ahead, block on the chain, then the next oldest ... -- the four waves of a SIMD end up running one at a
            // time (per-wave trace: half of every wave's loop time was hand-off wait) and a lone wave cannot fill the VALU.
            // A wave that is ahead of a SIMD mate (waves w, w+4, w+8, w+12 share a SIMD) drops to priority 0, the others
            // run at 3: the four stay within a phase of each other and interleave instruction by instruction.
#if MKB_HANDON_PRIO == 0  // (A/B builds, tools/kbench.py: 1 = no priority changes at all, 2 = a static priority by wave age, set once)
            int behind = 0x3fffffff;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                behind = min(behind, __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_done[(wave + 4 * k) & (NW - 1)], __ATOMIC_RELAXED,
                                                                                       __HIP_MEMORY_SCOPE_WORKGROUP)));
            if (finished > behind) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(3);
#endif
        };
#if MKB_HANDON_PRIO == 2
        if (wave >= 12) __builtin_amdgcn_s_setprio(3);
        else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
        if constexpr (DENSE) {
            // Dense pass.  In phase ph the wave owns chunk (wave + ph) mod 16 = lanes l0, l0 + cph, ... of half h; the first
            // nd = dense_lanes / cph of them are dense positions.  All eight rows take the pair body unconditionally (a row that
            // does not use the position carries seed 0 and adds exactly 0): straight-line code.
            //   seeds, pool ids: wave-uniform -> scalar cache (the 8 seeds of a tile are 32 contiguous bytes of the blocked
            //     layout), issued by hand at the top of a position's body and waited for at its end: left to itself the
            //     compiler sinks them in front of the s_waitcnt lgkmcnt(0) of the LDS read-modify-write -- a scalar-memory
            //     round trip per position;
            //   candidate rows: every wave of the workgroup walks the same 16 x nd positions, so the workgroup's slice of each
            //     row is brought into LDS ONCE, before the first tile (s_x, filled by the chunk's phase-0 owner), and read
            //     from there -- per-wave row loads were what the first version of this pass waited for (111 us; without the
            //     loads 83 us; a 3-deep DMA ring per wave changed nothing: it is the number of requests, not their latency).
            typedef float v8f __attribute__((ext_vector_type(8)));
            const int lcph = __builtin_ctz((unsigned)cph), nd = Ld >> lcph;  // (cph = 16 / halves is a power of two)
            const float *Gt = A.G + (((int64_t)tile * npb + pb) * halves) * 512;
            v8f gL = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto slot_of = [&](int ph_, int k_, int &p_, int &ds_) {
                const int c = (kChunkStride * wave + ph_) & (kChunks - 1);
                const int h_ = c >> lcph, j_ = (c & (cph - 1)) + (k_ << lcph);
                p_ = pb + npb * (j_ * halves + h_);
                ds_ = h_ * A.dense_lanes + j_;  // slot of the (dense-only) LDS accumulator
                return h_ * 64 + j_;            // slot of the block (seed layout, partial buffer)
            };
            auto step = [&](int &ph_, int &k_) {  // the position after (ph_, k_); the last one repeats (its loads are harmless)
                if (k_ + 1 < nd) { ++k_; }
                else if (ph_ + 1 < kChunks) { ++ph_; k_ = 0; }
            };
            auto uni = [](const void *q) {  // the address as a scalar register pair (every lane holds the same value)
                const unsigned long long v = (unsigned long long)q;
                return (const void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v));
            };
            auto sload_seeds = [&](int slot_) { asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(gL) : "s"(uni(Gt + slot_ * 8))); };
            // (the builtin tells the compiler's own wait-count bookkeeping that nothing is outstanding; the asm statement ties
            // the loaded registers to the wait)
            auto swait = [&]() {
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(gL)::"memory");
            };
            // (Round 4 looked at the ISA of this loop: per position the wave stops three times -- an s_waitcnt lgkmcnt(1) two
            // instructions into the body, lgkmcnt(0) before the accumulator add, lgkmcnt(0) behind the write.  Issuing the three LDS
            // operations by hand so that only the middle one remains was tried and dropped: the extra live registers pushed
            // the loop into scratch spills -- the kernel sits at 126 of 128 VGPRs -- and the first wait stayed.)
            acc_t xv_n = acc_t{};         // the next position's row image
            int slot_n = 0, dslot_n = 0;  // the current position's slots
            int ph1 = 0, k1 = 0;          // the position after the current one (its seeds are requested a position ahead)
            MKB_TRACE_ONLY(const unsigned long long td0 = __builtin_readcyclecounter(); if (t == 0) tr_pro = td0 - tr_c0;)
            if (nd > 0) {
                int p_;
                slot_n = slot_of(0, 0, p_, dslot_n);
                sload_seeds(slot_n);
                xv_n = s_x[(size_t)dslot_n * 64 + lane];
                swait();
                step(ph1, k1);
            }
            for (int ph = 0; ph < kChunks; ++ph) {
                for (int k = 0; k < nd; ++k) {
                    float x0[KPT], x1[KPT], g[TI];
                    {
                        const acc_t xv = xv_n;  // (read from LDS a position ago; lanes past the row's end hold 0)
                        if constexpr (NC == 1) { x0[0] = xv; x1[0] = 0.f; }
                        else if constexpr (NC == 2 && !CP) { x0[0] = xv.x; x0[1] = xv.y; x1[0] = 0.f; x1[1] = 0.f; }
                        else if constexpr (NC == 2) { x0[0] = xv.x; x1[0] = xv.y; }
                        else if constexpr (NC == 4 && !CP) {
                            x0[0] = xv.x; x0[1] = xv.y; x0[KPT - 2] = xv.z; x0[KPT - 1] = xv.w;
#pragma unroll
                            for (int v = 0; v < KPT; ++v) x1[v] = 0.f;
                        } else { x0[0] = xv.x; x0[1] = xv.y; x1[0] = xv.z; x1[1] = xv.w; }
                    }
#pragma unroll
                    for (int r = 0; r < TI; ++r) g[r] = gL[r];  // (rows past B: their seeds are written as 0 by the producers of G)
                    acc_t *slot = s_dx + (size_t)dslot_n * 64 + lane;
                    // requested now, taken over at the end of the body: the seeds of the next position
                    {
                        int p_;
                        slot_n = slot_of(ph1, k1, p_, dslot_n);
                        sload_seeds(slot_n);
                        xv_n = s_x[(size_t)dslot_n * 64 + lane];  // the next position's row image (both land under this body's math)
                        step(ph1, k1);
                        __builtin_amdgcn_sched_barrier(0);  // (or the scheduler sinks the address arithmetic -- and the load with it -- to the end of the body)
                    }
                    float dx0[KPT], dx1[KPT];
#pragma unroll
                    for (int v = 0; v < KPT; ++v) { dx0[v] = 0.f; dx1[v] = 0.f; }
                    if constexpr (CP && KPT == 2) {
#pragma unroll
                        for (int r = 0; r < TI; r += 2) {
                            f2 arA = f2{dq0[r][0], dq0[r][1]}, aiA = f2{dq1[r][0], dq1[r][1]};
                            f2 arB = f2{dq0[r + 1][0], dq0[r + 1][1]}, aiB = f2{dq1[r + 1][0], dq1[r + 1][1]};
                            f2 br = f2{dx0[0], dx0[1]}, bi = f2{dx1[0], dx1[1]};
                            pair_bwd_cmod2_both_x2(f2{q0[r][0], q0[r][1]}, f2{q1[r][0], q1[r][1]}, f2{q0[r + 1][0], q0[r + 1][1]},
                                                   f2{q1[r + 1][0], q1[r + 1][1]}, f2{x0[0], x0[1]}, f2{x1[0], x1[1]}, g[r], g[r + 1],
                                                   arA, aiA, arB, aiB, br, bi);
                            dq0[r][0] = arA.x; dq0[r][1] = arA.y; dq1[r][0] = aiA.x; dq1[r][1] = aiA.y;
                            dq0[r + 1][0] = arB.x; dq0[r + 1][1] = arB.y; dq1[r + 1][0] = aiB.x; dq1[r + 1][1] = aiB.y;
                            dx0[0] = br.x; dx0[1] = br.y; dx1[0] = bi.x; dx1[1] = bi.y;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < TI; ++r)
#pragma unroll
                            for (int v = 0; v < KPT; ++v) {
                                if constexpr (CP) {
                                    Cplx dq, dx;
                                    pair_bwd_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]}, g[r], dq, dx);
                                    dq0[r][v] += dq.re; dq1[r][v] += dq.im;
                                    dx0[v] += dx.re; dx1[v] += dx.im;
                                } else {
                                    float dq, dx, e0 = 0.f;
                                    pair_bwd_real<MODEL, HEAD>(q0[r][v], x0[v], g[r], A.kd, modulus, dq, dx, e0);
                                    dq0[r][v] += dq;
                                    dx0[v] += dx;
                                    extra += g[r] * e0;
                                }
                            }
                    }
                    acc_t upd = *slot;
                    if constexpr (NC == 1) upd += dx0[0];
                    else if constexpr (NC == 2 && !CP) { upd.x += dx0[0]; upd.y += dx0[1]; }
                    else if constexpr (NC == 2) { upd.x += dx0[0]; upd.y += dx1[0]; }
                    else if constexpr (NC == 4 && !CP) { upd.x += dx0[0]; upd.y += dx0[1]; upd.z += dx0[2]; upd.w += dx0[3]; }
                    else { upd.x += dx0[0]; upd.y += dx0[1]; upd.z += dx1[0]; upd.w += dx1[1]; }
                    *slot = upd;
                    swait();
                }
                hand_on(gd0 + ph + 1);
            }
            MKB_TRACE_ONLY(tr_dense += __builtin_readcyclecounter() - td0;)
        }
        // With a dense pass the sparse fringe (lanes >= dense_lanes) needs no chain at all: its dq half runs here, row-major, on
        // registers only (one stream of used fringe positions per half; the dx products of the shared bodies are dead code);
        // its dx half runs slot-major after the tile loop, every fringe slot owned by ONE wave of the workgroup.  (Sending the
        // fringe through the chunk rotation cost 88 k of the wave's 243 k cycles at the headline shape for 8 positions per
        // wave: every phase waited for whichever wave was building a run's stream.)
        if constexpr (DENSE) {
            MKB_TRACE_ONLY(const unsigned long long ts0 = __builtin_readcyclecounter();)
            // (the loads of the next half -- 8 seeds and the pool id of this lane's slot -- fly under the current half's work)
            float4 ga_n = make_float4(0.f, 0.f, 0.f, 0.f), gb_n = ga_n;
            int64_t id_nx = 0;
            auto half_loads = [&](int h_) {
                if (!have || h_ >= halves) return;
                const float4 *gp = reinterpret_cast<const float4 *>(A.G + ((((int64_t)tile * npb + pb) * halves + h_) * 64 + lane) * 8);
                ga_n = gp[0]; gb_n = gp[1];
                id_nx = A.pool[min(pb + npb * (lane * halves + h_), A.P - 1)];
            };
            half_loads(0);
            for (int h = 0; h < halves; ++h) {
                float gv[TI];
                int items = 0, off_lo = 0, off_hi = 0, total = 0;
                const float4 ga = ga_n, gb = gb_n;
                const int64_t id_own = id_nx;
                half_loads(h + 1);
                if (have) {  // this lane's slot of the half is (h, lane)
                    const int p_own = pb + npb * (lane * halves + h);
                    const bool valid = p_own < A.P;
                    unsigned nz = 0, pm = 0;
                    gv[0] = ga.x; gv[1] = ga.y; gv[2] = ga.z; gv[3] = ga.w; gv[4] = gb.x; gv[5] = gb.y; gv[6] = gb.z; gv[7] = gb.w;
#pragma unroll
                    for (int r = 0; r < TI; ++r) {
                        gv[r] = (valid && i0 + r < A.B) ? gv[r] : 0.f;
                        const unsigned u = __float_as_uint(gv[r]) << 1;
                        nz |= u;
                        pm |= u ? (0x10000u << (r >> 1)) : 0u;
                    }
                    const int64_t off = (valid ? id_own : 0) * A.De;
                    const bool part = nz != 0 && lane >= Ld;
                    const unsigned long long used_h = __ballot(nz != 0), pmask = __ballot(part);
                    if (lane == 0 && used_h) atomicOr(&s_used[h], used_h);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pmask, 0u));
                    total = __popcll(pmask);
                    const unsigned long long rest = ~pmask;
                    const int spare = __builtin_amdgcn_mbcnt_hi((unsigned)(rest >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rest, 0u));
                    const int dst = (part ? rank : total + spare) << 2;
                    items = __builtin_amdgcn_ds_permute(dst, (int)pm);
                    off_lo = __builtin_amdgcn_ds_permute(dst, (int)(unsigned)off);
                    off_hi = __builtin_amdgcn_ds_permute(dst, (int)(off >> 32));
#pragma unroll
                    for (int r = 0; r < TI; ++r)
                        gv[r] = __uint_as_float((unsigned)__builtin_amdgcn_ds_permute(dst, (int)__float_as_uint(gv[r])));
                }
                auto load_item = [&](int idx, float (&d0)[KPT], float (&d1)[KPT]) {
                    const int64_t off = ((int64_t)__builtin_amdgcn_readlane(off_hi, idx) << 32) |
                                        (int64_t)(unsigned)__builtin_amdgcn_readlane(off_lo, idx);
                    load_units<CP, KPT>(A.ent + off, A.d, NU, u0, d0, d1);
                };
                float xr0[kRingF][KPT], xr1[kRingF][KPT];
                const int last = total - 1;
                if (total > 0) {
#pragma unroll
                    for (int k = 0; k < kRingF; ++k) load_item(min(k, last), xr0[k], xr1[k]);
                }
                MKB_TRACE_ONLY(tr_items += total;)
                for (int it = 0; it < total; it += kRingF) {
#pragma unroll
                    for (int k = 0; k < kRingF; ++k) {
                        const int idx = it + k;
                        float x0[KPT], x1[KPT];
#pragma unroll
                        for (int v = 0; v < KPT; ++v) { x0[v] = xr0[k][v]; x1[v] = xr1[k][v]; }
                        load_item(min(idx + kRingF, last), xr0[k], xr1[k]);
                        if (idx >= total) continue;
                        const unsigned word = (unsigned)__builtin_amdgcn_readlane(items, idx);  // row-pair mask << 16
                        float dx0[KPT], dx1[KPT];  // (never read: the dx products are dead code here)
#pragma unroll
                        for (int v = 0; v < KPT; ++v) { dx0[v] = 0.f; dx1[v] = 0.f; }
#pragma unroll
                        for (int r = 0; r < TI; r += 2) {
                            if (!(word & (0x10000u << (r >> 1)))) continue;  // (scalar bit test: neither row of the pair uses the position)
                            const float gA = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(gv[r]), idx));
                            const float gB = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(gv[r + 1]), idx));
                            if constexpr (CP && KPT == 2) {
                                f2 arA = f2{dq0[r][0], dq0[r][1]}, aiA = f2{dq1[r][0], dq1[r][1]};
                                f2 arB = f2{dq0[r + 1][0], dq0[r + 1][1]}, aiB = f2{dq1[r + 1][0], dq1[r + 1][1]};
                                f2 br = f2{0.f, 0.f}, bi = f2{0.f, 0.f};
                                pair_bwd_cmod2_both_x2(f2{q0[r][0], q0[r][1]}, f2{q1[r][0], q1[r][1]}, f2{q0[r + 1][0], q0[r + 1][1]},
                                                       f2{q1[r + 1][0], q1[r + 1][1]}, f2{x0[0], x0[1]}, f2{x1[0], x1[1]}, gA, gB,
                                                       arA, aiA, arB, aiB, br, bi);
                                dq0[r][0] = arA.x; dq0[r][1] = arA.y; dq1[r][0] = aiA.x; dq1[r][1] = aiA.y;
                                dq0[r + 1][0] = arB.x; dq0[r + 1][1] = arB.y; dq1[r + 1][0] = aiB.x; dq1[r + 1][1] = aiB.y;
                            } else {
#pragma unroll
                                for (int rr = r; rr < r + 2; ++rr) {
                                    const float gr = rr == r ? gA : gB;
#pragma unroll
                                    for (int v = 0; v < KPT; ++v) {
                                        if constexpr (CP) {
                                            Cplx dq, dx;
                                            pair_bwd_cmod(Cplx{q0[rr][v], q1[rr][v]}, Cplx{x0[v], x1[v]}, gr, dq, dx);
                                            dq0[rr][v] += dq.re; dq1[rr][v] += dq.im;
                                        } else {
                                            float dq, dx, e0 = 0.f;
                                            pair_bwd_real<MODEL, HEAD>(q0[rr][v], x0[v], gr, A.kd, modulus, dq, dx, e0);
                                            dq0[rr][v] += dq;
                                            extra += gr * e0;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // (the ring's last requests are never consumed: were they still pending -- in the compiler's bookkeeping -- when the
            // tile loop comes round, it would guard the registers they target with vmcnt waits inside the dense loop, where
            // they drain the DMA ring it cannot see)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            MKB_TRACE_ONLY(tr_setup += __builtin_readcyclecounter() - ts0;)
        } else
        // The 16 phases of a tile split into RUNS of consecutive phases whose chunks lie in the same half (the seeds and ids
        // of one half fit the lanes).  Inside a run the wave's used positions form one stream, ordered by phase: everything
        // the loop needs per position is PERMUTED INTO STREAM ORDER once per run (lane i = i-th position: its slot, phase,
        // row-pair mask, table offset, 8 seeds), so the loop body fetches it with v_readlane at a scalar index and spends
        // ~25 scalar instructions per position.  (The scalar unit is shared by the CU's 16 waves: an earlier version that
        // walked the bit masks in the loop -- ~150 scalar instructions per position -- was bound by it, the pair math did
        // not matter.)  Candidate rows are requested kRing1 positions ahead, across phase boundaries: x is read-only, only
        // the LDS update has to wait for its phase.
        for (int ph0 = 0; ph0 < kChunks;) {
            const int c0 = (kChunkStride * wave + ph0) & (kChunks - 1);
            const int h = c0 / cph, l00 = c0 - h * cph;
            const int len = min(kChunks - ph0, cph - l00), ph1 = ph0 + len;  // phases [ph0, ph1): chunk lane offsets l00, l00 + 1, ...
            MKB_TRACE_ONLY(const unsigned long long ts0 = __builtin_readcyclecounter();)
            float gv[TI];
            int items = 0, off_lo = 0, off_hi = 0, total = 0;
            if (have) {  // this lane's slot of the half is (h, lane)
                const int p_own = pb + npb * (lane * halves + h);
                const bool valid = p_own < A.P;
                unsigned nz = 0, pm = 0;
                if (A.g_blocked) {  // the lane's 8 seeds are contiguous: two 16-byte loads
                    const float4 *gp = reinterpret_cast<const float4 *>(A.G + ((((int64_t)tile * npb + pb) * halves + h) * 64 + lane) * 8);
                    const float4 ga = gp[0], gb = gp[1];
                    gv[0] = ga.x; gv[1] = ga.y; gv[2] = ga.z; gv[3] = ga.w; gv[4] = gb.x; gv[5] = gb.y; gv[6] = gb.z; gv[7] = gb.w;
                }
#pragma unroll
                for (int r = 0; r < TI; ++r) {
                    if (A.g_blocked) gv[r] = (valid && i0 + r < A.B) ? gv[r] : 0.f;
                    else gv[r] = (valid && i0 + r < A.B) ? A.G[(int64_t)(i0 + r) * A.P + p_own] : 0.f;
                    const unsigned u = __float_as_uint(gv[r]) << 1;  // +-0 -> unused pair (the loss kernel writes 0 for them)
                    nz |= u;
                    pm |= u ? (0x10000u << (r >> 1)) : 0u;
                }
                const int64_t off = (valid ? A.pool[p_own] : 0) * A.De;
                const int rp = (lane % cph) - l00;  // phase of this lane's chunk, relative to the run
                const bool part = nz != 0 && rp >= 0 && rp < len && lane >= Ld;  // (dense lanes were done by the dense pass)
                const unsigned long long used_h = __ballot(nz != 0), pmask = __ballot(part);
                if (lane == 0 && used_h) atomicOr(&s_used[h], used_h);
                // stream rank of this lane's position: positions of earlier phases + earlier lanes of the same chunk
                int rank = 0, pre = 0;
                for (int q = 0; q < len; ++q) {
                    const unsigned long long mq = pmask & (cm << (l00 + q));
                    const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mq, 0u));
                    rank = (rp == q) ? pre + below : rank;
                    pre += __popcll(mq);
                }
                total = pre;
                const unsigned long long rest = ~pmask;  // the other lanes take the ranks behind the stream: a full permutation
                const int spare = __builtin_amdgcn_mbcnt_hi((unsigned)(rest >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rest, 0u));
                const int dst = (part ? rank : total + spare) << 2;
                items = __builtin_amdgcn_ds_permute(dst, (int)(pm | ((unsigned)rp << 8) | (unsigned)lane));
                off_lo = __builtin_amdgcn_ds_permute(dst, (int)(unsigned)off);
                off_hi = __builtin_amdgcn_ds_permute(dst, (int)(off >> 32));
#pragma unroll
                for (int r = 0; r < TI; ++r)
                    gv[r] = __uint_as_float((unsigned)__builtin_amdgcn_ds_permute(dst, (int)__float_as_uint(gv[r])));
            } else {
#pragma unroll
                for (int r = 0; r < TI; ++r) gv[r] = 0.f;
            }
            auto load_item = [&](int idx, float (&d0)[KPT], float (&d1)[KPT]) {
                const int64_t off = ((int64_t)__builtin_amdgcn_readlane(off_hi, idx) << 32) |
                                    (int64_t)(unsigned)__builtin_amdgcn_readlane(off_lo, idx);
                load_units<CP, KPT>(A.ent + off, A.d, NU, u0, d0, d1);
            };
            float xr0[kRing1][KPT], xr1[kRing1][KPT];
            const int last = total - 1;
            if (total > 0) {
#pragma unroll
                for (int k = 0; k < kRing1; ++k) load_item(min(k, last), xr0[k], xr1[k]);
            }
            MKB_TRACE_ONLY(tr_setup += __builtin_readcyclecounter() - ts0; tr_items += total;)
            int cur = ph0;  // barriers passed = phase this wave is in
            for (int it = 0; it < total; it += kRing1) {
#pragma unroll
                for (int k = 0; k < kRing1; ++k) {
                    const int idx = it + k;
                    float x0[KPT], x1[KPT];
#pragma unroll
                    for (int v = 0; v < KPT; ++v) { x0[v] = xr0[k][v]; x1[v] = xr1[k][v]; }
                    load_item(min(idx + kRing1, last), xr0[k], xr1[k]);
                    if (idx >= total) continue;
                    const unsigned word = (unsigned)__builtin_amdgcn_readlane(items, idx);  // slot | phase << 8 | pair mask << 16
                    const int j = (int)(word & 63u), jph = ph0 + (int)((word >> 8) & 63u);
                    // hand the chunks on: LDS writes done, then the barrier (global loads stay in flight)
                    for (; cur < jph; ++cur) hand_on(g0 + cur + 1);
                    acc_t *slot = s_dx + (size_t)(h * 64 + j) * 64 + lane;
                    float dx0[KPT], dx1[KPT];
#pragma unroll
                    for (int v = 0; v < KPT; ++v) { dx0[v] = 0.f; dx1[v] = 0.f; }
                    float g[TI];
#pragma unroll
                    for (int r = 0; r < TI; ++r)
                        g[r] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(gv[r]), idx));
                    auto one_row = [&](int r) {
                        if constexpr (CP && KPT % 2 == 0) {
#pragma unroll
                            for (int v = 0; v < KPT; v += 2) {
                                f2 ar = f2{dq0[r][v], dq0[r][v + 1]}, ai = f2{dq1[r][v], dq1[r][v + 1]};
                                f2 br = f2{dx0[v], dx0[v + 1]}, bi = f2{dx1[v], dx1[v + 1]};
                                pair_bwd_cmod2_both(f2{q0[r][v], q0[r][v + 1]}, f2{q1[r][v], q1[r][v + 1]},
                                                    f2{x0[v], x0[v + 1]}, f2{x1[v], x1[v + 1]}, g[r], ar, ai, br, bi);
                                dq0[r][v] = ar.x; dq0[r][v + 1] = ar.y;
                                dq1[r][v] = ai.x; dq1[r][v + 1] = ai.y;
                                dx0[v] = br.x; dx0[v + 1] = br.y;
                                dx1[v] = bi.x; dx1[v + 1] = bi.y;
                            }
                        } else
#pragma unroll
                        for (int v = 0; v < KPT; ++v) {
                            if constexpr (CP) {
                                Cplx dq, dx;
                                pair_bwd_cmod(Cplx{q0[r][v], q1[r][v]}, Cplx{x0[v], x1[v]}, g[r], dq, dx);
                                dq0[r][v] += dq.re; dq1[r][v] += dq.im;
                                dx0[v] += dx.re; dx1[v] += dx.im;
                            } else {
                                float dq, dx, e0 = 0.f;
                                pair_bwd_real<MODEL, HEAD>(q0[r][v], x0[v], g[r], A.kd, modulus, dq, dx, e0);
                                dq0[r][v] += dq;
                                dx0[v] += dx;
                                extra += g[r] * e0;
                            }
                        }
                    };
#pragma unroll
                    for (int r = 0; r < TI; r += 2) {
                        if constexpr (CP && KPT == 2) {
                            // Rows go two at a time: two interleaved dependent chains fill each other's idle issue slots.  A row
                            // of the pair that does not use the position carries g = 0 and adds exactly 0 (one body per pair
                            // keeps the accumulators in place; alternative bodies cost register moves at their join).
                            if (word & (0x10000u << (r >> 1))) {  // (scalar bit test: some row of the pair uses the position)
                                f2 arA = f2{dq0[r][0], dq0[r][1]}, aiA = f2{dq1[r][0], dq1[r][1]};
                                f2 arB = f2{dq0[r + 1][0], dq0[r + 1][1]}, aiB = f2{dq1[r + 1][0], dq1[r + 1][1]};
                                f2 br = f2{dx0[0], dx0[1]}, bi = f2{dx1[0], dx1[1]};
                                pair_bwd_cmod2_both_x2(f2{q0[r][0], q0[r][1]}, f2{q1[r][0], q1[r][1]}, f2{q0[r + 1][0], q0[r + 1][1]},
                                                       f2{q1[r + 1][0], q1[r + 1][1]}, f2{x0[0], x0[1]}, f2{x1[0], x1[1]}, g[r], g[r + 1],
                                                       arA, aiA, arB, aiB, br, bi);
                                dq0[r][0] = arA.x; dq0[r][1] = arA.y; dq1[r][0] = aiA.x; dq1[r][1] = aiA.y;
                                dq0[r + 1][0] = arB.x; dq0[r + 1][1] = arB.y; dq1[r + 1][0] = aiB.x; dq1[r + 1][1] = aiB.y;
                                dx0[0] = br.x; dx0[1] = br.y; dx1[0] = bi.x; dx1[1] = bi.y;
                            }
                            continue;
                        }
                        if ((__float_as_uint(g[r]) << 1) != 0u) one_row(r);
                        if ((__float_as_uint(g[r + 1]) << 1) != 0u) one_row(r + 1);
                    }
                    // this wave owns the chunk during the phase: plain read-modify-write (read late: 4 VGPRs less across the
                    // rows).  Component order: [re/real KPT][im KPT]
                    acc_t upd = *slot;
                    if constexpr (NC == 1) upd += dx0[0];
                    else if constexpr (NC == 2 && !CP) { upd.x += dx0[0]; upd.y += dx0[1]; }
                    else if constexpr (NC == 2) { upd.x += dx0[0]; upd.y += dx1[0]; }
                    else if constexpr (NC == 4 && !CP) { upd.x += dx0[0]; upd.y += dx0[1]; upd.z += dx0[2]; upd.w += dx0[3]; }
                    else { upd.x += dx0[0]; upd.y += dx0[1]; upd.z += dx1[0]; upd.w += dx1[1]; }
                    *slot = upd;
                }
            }
            for (; cur < ph1; ++cur) hand_on(g0 + cur + 1);
            ph0 = ph1;
        }
        if (have) {
            float *dQs = A.dQ + (int64_t)pb * A.B * A.De;
#pragma unroll
            for (int r = 0; r < TI; ++r)
                if (i0 + r < A.B) store_units<CP, KPT>(dQs + (int64_t)(i0 + r) * A.De, A.d, NU, u0, dq0[r], dq1[r]);
            if constexpr (MODEL == MKB_PROTATE) {  // d score / d modulus = - sum_k |sin z|   (protate.py:91)
                extra = wave_sum(extra);
                if (lane == 0) atomicAdd(A.g_modulus, -extra);
            }
        }
        if constexpr (DENSE) {
            if (A.tiles_per_wave == 1) {  // (workgroup-uniform; every wave runs the tile loop exactly once)
                // The fringe's dx half (below) walks, per fringe slot, the rows of the WORKGROUP that use it.  Fetched from the
                // Q buffer those rows were what that part waited for (16 scattered 1 KB requests per batch: 47-64 k cycles on the
                // wave with the heaviest slot); the rows are in registers right here, and LDS is free once the accumulator has
                // been flushed: flush, then park the tile's eight rows for the slot owners.
                __syncthreads();  // every wave's dense pass is over
                {
                    const size_t wg = (size_t)rg * npb + pb;
                    acc_t *out = reinterpret_cast<acc_t *>(A.dXp) + ((wg * cap) * A.dim_slices + s) * 64 + lane;
                    for (int sidx = wave; sidx < cap; sidx += NW) {
                        if ((sidx & 63) >= A.dense_lanes || !((s_used[sidx >> 6] >> (sidx & 63)) & 1ull)) continue;
                        out[(size_t)sidx * A.dim_slices * 64] = s_dx[(size_t)((sidx >> 6) * A.dense_lanes + (sidx & 63)) * 64 + lane];
                    }
                }
                __syncthreads();  // the accumulator has been read out
#pragma unroll
                for (int r = 0; r < TI; ++r) {
                    acc_t v;
                    if constexpr (NC == 1) v = q0[r][0];
                    else if constexpr (NC == 2 && !CP) { v.x = q0[r][0]; v.y = q0[r][1]; }
                    else if constexpr (NC == 2) { v.x = q0[r][0]; v.y = q1[r][0]; }
                    else if constexpr (NC == 4 && !CP) { v.x = q0[r][0]; v.y = q0[r][1]; v.z = q0[r][KPT - 2]; v.w = q0[r][KPT - 1]; }
                    else { v.x = q0[r][0]; v.y = q0[r][1]; v.z = q1[r][0]; v.w = q1[r][1]; }
                    s_q[(size_t)(wave * TI + r) * 64 + lane] = v;  // (a wave without a tile parks zeros)
                }
            }
        }
    }
    if constexpr (DENSE) {
        // dx of the fringe, slot-major.  The used-slot masks are complete once every wave has left its tile loop (barrier);
        // the used fringe slots are dealt round-robin to the 16 waves (in pool order: the rows per slot fall with the position,
        // so the deal is balanced).  The owner walks the rows of the workgroup that use the slot -- the seeds of 128 rows are 8
        // bytes per lane of the blocked layout -- sixteen rows at a time (their query rows come from the Q buffer: 32 loads in
        // flight), sums their dx in registers in a fixed order and STORES the slot's partial: nobody else touches it.  This
        // part is all latency, so everything a wave's slots need first (seeds, pool ids, then candidate rows) is requested
        // for up to four slots at once.
        __syncthreads();
        MKB_TRACE_ONLY(const unsigned long long tf0 = __builtin_readcyclecounter();)
        constexpr int NB = 4;  // rows per batch: SMALL on purpose -- this code runs once per launch, and every 64 bytes of it are an instruction-cache miss the first time (a 16-row batch, unrolled, cost more in cold code than it saved in round trips)
        const int row_tiles2 = (A.B + TI - 1) / TI;
        const unsigned long long fmask = A.dense_lanes >= 64 ? 0ull : ~0ull << A.dense_lanes;
        int n_used = 0;
        for (int h = 0; h < halves; ++h) n_used += __popcll(s_used[h] & fmask);
        n_used = __builtin_amdgcn_readfirstlane(n_used);
        auto nth_slot = [&](int r, int &h_, int &j_) {  // the r-th used fringe slot in pool order (r < n_used)
            for (int h = 0; h < halves; ++h) {
                unsigned long long bits = s_used[h] & fmask;
                bits = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) |
                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)bits);
                const int c = __popcll(bits);
                if (r >= c) { r -= c; continue; }
                for (; r > 0; --r) bits &= bits - 1ull;
                h_ = h; j_ = (int)__builtin_ctzll(bits);
                return;
            }
            h_ = 0; j_ = 0;
        };
        // this wave's slots: ranks wave, wave + 16, ...; the seeds (of the first 128 rows) and the pool id of the NEXT slot are
        // requested while the current one is worked on, its candidate row as soon as the id is in
        const int tile_l0 = rg * A.tiles_per_wave * NW + (lane >> 2);
        float n_gsx = 0.f, n_gsy = 0.f, nx0[KPT], nx1[KPT];
        int n_h = 0, n_j = 0;
        int64_t n_id = 0;
        auto request = [&](int r) {  // seeds and candidate row of slot rank r (nothing is waited for here)
            if (r >= n_used) return;
            nth_slot(r, n_h, n_j);
            n_id = s_ids[n_h * 64 + n_j];
            load_units_raw<CP, KPT>(A.ent + n_id * A.De, A.d, NU, u0, nx0, nx1);  // (lanes past the row's end: never read back)
            n_gsx = 0.f; n_gsy = 0.f;
            if (tile_l0 < row_tiles2) {
                const float2 gs = *reinterpret_cast<const float2 *>(A.G + ((((int64_t)tile_l0 * npb + pb) * halves + n_h) * 64 + n_j) * 8 + (lane & 3) * 2);
                n_gsx = gs.x; n_gsy = gs.y;
            }
        };
#pragma unroll
        for (int v = 0; v < KPT; ++v) { nx0[v] = 0.f; nx1[v] = 0.f; }
        // the deal goes back and forth (ranks w, 31 - w, 32 + w, ...): the rows per slot fall with the rank, so the wave with
        // the heaviest slot of one round gets the lightest of the next.  (Round 6 tried a DYNAMIC deal -- one LDS counter, the
        // next rank claimed a slot ahead -- against the slowest wave's 21.5 k cycles here (mean 9.5 k, head-batch): slower,
        // 93.7 / 87.0 us against 91.5 / 85.7 (head / tail, steady state, two repetitions in one call).  Removed.
        // Also tried and removed in round 6: keeping the dx products of the fringe's row-major half (they are evaluated there
        // anyway, rows in registers) and adding them to LDS accumulators with ds_add_f32 -- no second half at all, every parity
        // test green, and 133 / 95.6 us against 91.6 / 85.7: a wave's ds_add_f32 costs thousands of cycles here, not the 128
        // the dense experiment of round 2 suggested.  profiles/r06_ab_experiments.txt)
        auto rank_of = [&](int k) { return k * NW + ((k & 1) ? NW - 1 - wave : wave); };
        request(rank_of(0));
        for (int k = 0, r = rank_of(0); k * NW < n_used; ++k, r = rank_of(k)) {
            if (r >= n_used) continue;  // (only in the last round; nothing was requested for it)
            const int r_next = (k + 1) * NW < n_used ? rank_of(k + 1) : n_used;
            const int j = n_j, h = n_h;
            float x0[KPT], x1[KPT], dx0[KPT], dx1[KPT];
            const float gsx = n_gsx, gsy = n_gsy;
#pragma unroll
            for (int v = 0; v < KPT; ++v) { x0[v] = nx0[v]; x1[v] = nx1[v]; dx0[v] = 0.f; dx1[v] = 0.f; }
            float ax0[4][KPT], ax1[4][KPT];
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
                for (int v = 0; v < KPT; ++v) { ax0[a4][v] = 0.f; ax1[a4][v] = 0.f; }
            request(r_next);
            for (int t2 = 0; t2 < A.tiles_per_wave; ++t2) {
                const int tile_b = (rg * A.tiles_per_wave + t2) * NW;  // 16 tiles = 128 rows; lane -> rows 2 * lane, 2 * lane + 1 of them
                const int tile_l = tile_b + (lane >> 2), ib = tile_b * TI;
                float vx = gsx, vy = gsy;
                if (t2 > 0) {
                    vx = 0.f; vy = 0.f;
                    if (tile_l < row_tiles2) {
                        const float2 gs = *reinterpret_cast<const float2 *>(A.G + ((((int64_t)tile_l * npb + pb) * halves + h) * 64 + j) * 8 + (lane & 3) * 2);
                        vx = gs.x; vy = gs.y;
                    }
                }
                vx = ib + 2 * lane < A.B ? vx : 0.f;
                vy = ib + 2 * lane + 1 < A.B ? vy : 0.f;
                // rows in a fixed order: the even rows of the 128 (lane order), then the odd ones; sixteen rows per batch, their
                // dx products on FOUR accumulators that take turns (one accumulator chains all the bodies of a batch: a lone
                // wave -- few waves have a heavy slot -- then runs at the latency of every packed op) and are added up in a fixed
                // order at the end
                const unsigned long long me = __ballot((__float_as_uint(vx) << 1) != 0u), mo = __ballot((__float_as_uint(vy) << 1) != 0u);
                MKB_TRACE_ONLY(tr_p2rows += __popcll(me) + __popcll(mo); tr_p2slots += 1; const unsigned long long tb0 = __builtin_readcyclecounter();)
                for (int pass = 0; pass < 2; ++pass) {
                    unsigned long long m = pass ? mo : me;  // (wave-uniform)
                    const float vv = pass ? vy : vx;
                    while (m) {
                        float gr[NB], qa0[NB][KPT], qa1[NB][KPT];
                        int li[NB];  // row of the 128 (2 * lane + pass)
#pragma unroll
                        for (int c = 0; c < NB; ++c) {  // the next sixteen rows (short of sixteen: row 0 with seed 0)
                            const bool ok = m != 0ull;
                            const int L = ok ? (int)__builtin_ctzll(m) : 0;
                            const float gl = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(vv), L));
                            gr[c] = ok ? gl : 0.f;
                            m &= m - 1ull;  // (x & (x - 1) of 0 is 0)
                            li[c] = 2 * L + pass;
                        }
                        if (A.tiles_per_wave == 1) {  // parked in LDS by the rows' owners: sixteen reads in flight
#pragma unroll
                            for (int c = 0; c < NB; ++c) {
                                const acc_t qv = s_q[(size_t)li[c] * 64 + lane];
                                if constexpr (NC == 1) { qa0[c][0] = qv; qa1[c][0] = 0.f; }
                                else if constexpr (NC == 2 && !CP) { qa0[c][0] = qv.x; qa0[c][1] = qv.y; qa1[c][0] = 0.f; qa1[c][1] = 0.f; }
                                else if constexpr (NC == 2) { qa0[c][0] = qv.x; qa1[c][0] = qv.y; }
                                else if constexpr (NC == 4 && !CP) {
                                    qa0[c][0] = qv.x; qa0[c][1] = qv.y; qa0[c][KPT - 2] = qv.z; qa0[c][KPT - 1] = qv.w;
#pragma unroll
                                    for (int v = 0; v < KPT; ++v) qa1[c][v] = 0.f;
                                } else { qa0[c][0] = qv.x; qa0[c][1] = qv.y; qa1[c][0] = qv.z; qa1[c][1] = qv.w; }
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < NB; ++c)
                                load_units_raw<CP, KPT>(A.Q + (int64_t)min(ib + li[c], A.B - 1) * A.De, A.d, NU, u0, qa0[c], qa1[c]);
                        }
#pragma unroll
                        for (int c = 0; c < NB; c += 2) {
                            constexpr int NA = 4;
                            const int a4 = (c >> 1) & (NA - 1);
                            if constexpr (CP && KPT == 2) {
                                f2 arA = f2{0.f, 0.f}, aiA = arA, arB = arA, aiB = arA;  // (the dq products are dead code here)
                                f2 br = f2{ax0[a4][0], ax0[a4][1]}, bi = f2{ax1[a4][0], ax1[a4][1]};
                                pair_bwd_cmod2_both_x2(f2{qa0[c][0], qa0[c][1]}, f2{qa1[c][0], qa1[c][1]}, f2{qa0[c + 1][0], qa0[c + 1][1]},
                                                       f2{qa1[c + 1][0], qa1[c + 1][1]}, f2{x0[0], x0[1]}, f2{x1[0], x1[1]},
                                                       gr[c], gr[c + 1], arA, aiA, arB, aiB, br, bi);
                                ax0[a4][0] = br.x; ax0[a4][1] = br.y; ax1[a4][0] = bi.x; ax1[a4][1] = bi.y;
                            } else {
#pragma unroll
                                for (int cc = c; cc < c + 2; ++cc)
#pragma unroll
                                    for (int v = 0; v < KPT; ++v) {
                                        if constexpr (CP) {
                                            Cplx dq, dx;
                                            pair_bwd_cmod(Cplx{qa0[cc][v], qa1[cc][v]}, Cplx{x0[v], x1[v]}, gr[cc], dq, dx);
                                            ax0[a4][v] += dx.re; ax1[a4][v] += dx.im;
                                        } else {
                                            float dq, dx, e0 = 0.f;
                                            pair_bwd_real<MODEL, HEAD>(qa0[cc][v], x0[v], gr[cc], A.kd, modulus, dq, dx, e0);
                                            ax0[a4][v] += dx;
                                        }
                                    }
                            }
                        }
                    }
                }
                MKB_TRACE_ONLY(tr_p2batch += __builtin_readcyclecounter() - tb0;)
            }
#pragma unroll
            for (int v = 0; v < KPT; ++v) {
                dx0[v] = (ax0[0][v] + ax0[1][v]) + (ax0[2][v] + ax0[3][v]);
                dx1[v] = (ax1[0][v] + ax1[1][v]) + (ax1[2][v] + ax1[3][v]);
            }
            acc_t upd;
            if constexpr (NC == 1) upd = dx0[0];
            else if constexpr (NC == 2 && !CP) { upd.x = dx0[0]; upd.y = dx0[1]; }
            else if constexpr (NC == 2) { upd.x = dx0[0]; upd.y = dx1[0]; }
            else if constexpr (NC == 4 && !CP) { upd.x = dx0[0]; upd.y = dx0[1]; upd.z = dx0[2]; upd.w = dx0[3]; }
            else { upd.x = dx0[0]; upd.y = dx0[1]; upd.z = dx1[0]; upd.w = dx1[1]; }
            reinterpret_cast<acc_t *>(A.dXp)[((((size_t)rg * npb + pb) * cap + (h * 64 + j)) * A.dim_slices + s) * 64 + lane] = upd;
        }
        MKB_TRACE_ONLY(tr_p2 += __builtin_readcyclecounter() - tf0;)
    }
    MKB_TRACE_T(tr_t2);
#ifdef MKB_TRACE_WG
    if (lane == 0 && A.trace && A.trace_kind == 4) {  // per-wave record (tools/wgtrace.py run bwd1): cycles
        unsigned long long *tr = A.trace + 8ull * ((unsigned long long)blockIdx.x * NW + wave);
        tr[0] = __builtin_readcyclecounter() - tr_c0; tr[1] = tr_hand; tr[2] = tr_setup; tr[3] = 1 | (tr_p2slots << 1) | (tr_p2rows << 8) | (tr_p2batch << 24); tr[4] = tr_items | (tr_pro << 16); tr[5] = wave; tr[6] = tr_dense; tr[7] = tr_p2;
    }
#endif
    __syncthreads();  // every wave's last phase is done
    // LDS accumulator -> this row group's partial buffer (16 B per lane, 1 KB per slot and store: plain coalesced stores).
    // pool_dx_reduce_kernel sums the row groups and adds the result to the table gradient rows: one fp32 atomic per element
    // instead of one per element AND row group (8.4 M atomics at the headline shape, all issued when the workgroups finish
    // together: 26-37 us of the launch).
    {
        const size_t wg = (size_t)rg * npb + pb;
        if (s == 0 && tid < 8) A.xused[wg * 8 + tid] = tid < halves ? s_used[tid] : 0ull;
        acc_t *out = reinterpret_cast<acc_t *>(A.dXp) + ((wg * cap) * A.dim_slices + s) * 64 + lane;
        for (int sidx = wave; sidx < cap; sidx += NW) {
            if (!((s_used[sidx >> 6] >> (sidx & 63)) & 1ull)) continue;
            if constexpr (DENSE) {  // (the fringe slots went straight to the buffer; with one tile per wave the dense ones as well)
                if ((sidx & 63) >= A.dense_lanes || A.tiles_per_wave == 1) continue;
                out[(size_t)sidx * A.dim_slices * 64] = s_dx[(size_t)((sidx >> 6) * A.dense_lanes + (sidx & 63)) * 64 + lane];
            } else {
                out[(size_t)sidx * A.dim_slices * 64] = s_dx[(size_t)sidx * 64 + lane];
            }
        }
    }
    MKB_TRACE_OUT(A, 1, tr_t0, tr_t1, tr_t2, 0);
}

// Sums the row groups' dx partials of one slot and adds them to the slot's table gradient row.  One 256-lane workgroup per
// (block, slot); fixed summation order (row group 0, 1, ...): the only atomics left are the final adds (pool duplicates
// and the positive triples' rows share gradient rows).  Bandwidth-bound, so kpt / complex layout are run-time values:
// the body is shared by the stand-alone kernel and by the row backward kernel, whose launch it rides in the fused step.
struct DxReduce {
    const float *dXp;
    const unsigned long long *xused;
    const int64_t *pool;
    float *g_ent;
    int64_t De;
    int P, d, npb, halves, dim_slices, row_groups, kpt, cplx;
    int blocks;  // npb * halves * 64
    const int *occ;  // rider of the row backward: occurrence counts of the batch's entities (RowStepArgs::occ), or null
    int clear;       // the gradient rows are known to be all-zero (mkb_grads_t::rows_clear): exclusive rows are stored, not added to
};

__device__ __forceinline__ void pool_dx_reduce_block(const DxReduce &R, int block) {
    const int cap = R.halves * 64;
    const int pb = block / cap, sidx = block - pb * cap;
    const int h = sidx >> 6, l = sidx & 63;
    const int p = pb + R.npb * (l * R.halves + h);
    if (p >= R.P) return;
    // which row groups used the slot: lane rg reads the mask of row group rg (all loads at once), ballots collect them
    // (used_rg / used_rg2: row groups 0-63 / 64-127; `more`: somebody beyond those -- the plain loop at the bottom re-reads)
    unsigned long long used_rg = 0ull, used_rg2 = 0ull;
    bool more = false;
    for (int rg0 = 0; rg0 < R.row_groups; rg0 += 64) {
        const int rg = rg0 + (int)(threadIdx.x & 63);
        const bool u = rg < R.row_groups && ((R.xused[((size_t)rg * R.npb + pb) * 8 + h] >> l) & 1ull) != 0ull;
        const unsigned long long b = __ballot(u);
        if (rg0 == 0) used_rg = b;
        else if (rg0 == 64) used_rg2 = b;
        else more |= b != 0ull;
    }
    if (!used_rg && !used_rg2 && !more) return;
    const int nc = R.kpt * (R.cplx ? 2 : 1), NU = R.cplx ? R.d : (int)R.De;
    const int64_t ent = R.pool[p];
    float *row = R.g_ent + ent * R.De;
    // an entity that occurs once among the batch's pool ids, heads and tails: nobody else writes its gradient row in this
    // launch -> plain read-modify-write (workgroup-uniform), or a plain store when the row is known to be zero
    const bool own = R.occ && R.occ[ent] == 1;
    const bool store = own && R.clear;
    auto add = [&](float *dst, float v) { if (store) *dst = v; else if (own) *dst += v; else atomicAdd(dst, v); };
    const int per_slot = R.dim_slices * 64 * nc;  // floats of one slot of one row group
    if (nc == 4 && R.cplx && R.row_groups <= 128) {  // RotatE with two complex dims per lane: 16-byte loads, [re0 re1 im0 im1] per lane
        for (int e = threadIdx.x; e < R.dim_slices * 64; e += 256) {
            const int u = e * 2;
            if (u >= NU) continue;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            // eight row groups at a time: their partial rows are requested together (a loop with a run-time trip count and a
            // skip inside made every row group its own round trip), added in row-group order.  The "used" bits come from the two
            // ballots above and from nowhere else: with a re-read of xused as the fallback for later row groups inside this
            // loop, the compiler issued that load for EVERY row group and waited for it before the row's own load (round 5, ISA:
            // sixteen serial round trips where one was meant)
            for (int rg0 = 0; rg0 < R.row_groups; rg0 += 8) {
                float4 v[8];
                bool on[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int rg = rg0 + k;
                    on[k] = rg < R.row_groups && (((rg < 64 ? used_rg : used_rg2) >> (rg & 63)) & 1ull) != 0ull;
                    const int rgc = on[k] ? rg : 0;  // (row group 0's partial buffer always exists: a harmless address)
                    v[k] = *reinterpret_cast<const float4 *>(R.dXp + (((size_t)rgc * R.npb + pb) * cap + sidx) * per_slot + 4 * e);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (on[k]) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
            }
            // (u is even and NU = d a multiple of 2 on this path: [re0 re1] and [im0 im1] are 8-byte pairs)
            if (own) {
                float2 *pr = reinterpret_cast<float2 *>(row + u), *pi = reinterpret_cast<float2 *>(row + R.d + u);
                if (store) { *pr = make_float2(a.x, a.y); *pi = make_float2(a.z, a.w); }
                else {
                    const float2 o0 = *pr, o1 = *pi;
                    *pr = make_float2(o0.x + a.x, o0.y + a.y);
                    *pi = make_float2(o1.x + a.z, o1.y + a.w);
                }
            } else {
                atomicAdd(row + u, a.x); atomicAdd(row + u + 1, a.y);
                atomicAdd(row + R.d + u, a.z); atomicAdd(row + R.d + u + 1, a.w);
            }
        }
        return;
    }
    if (nc == 4 && !R.cplx && R.row_groups <= 128 && (R.De & 3) == 0) {  // four real units per lane (TransE, hidden % 256 == 0 or not): 16-byte loads
        for (int e = threadIdx.x; e < R.dim_slices * 64; e += 256) {
            const int u = e * 4;
            if (u >= NU) continue;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int rg0 = 0; rg0 < R.row_groups; rg0 += 8) {  // as above: eight row groups' partial rows together
                float4 v[8];
                bool on[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int rg = rg0 + k;
                    on[k] = rg < R.row_groups && (((rg < 64 ? used_rg : used_rg2) >> (rg & 63)) & 1ull) != 0ull;
                    const int rgc = on[k] ? rg : 0;
                    v[k] = *reinterpret_cast<const float4 *>(R.dXp + (((size_t)rgc * R.npb + pb) * cap + sidx) * per_slot + 4 * e);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (on[k]) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
            }
            float4 *pr = reinterpret_cast<float4 *>(row + u);
            if (store) *pr = a;
            else if (own) { const float4 o = *pr; *pr = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w); }
            else { atomicAdd(row + u, a.x); atomicAdd(row + u + 1, a.y); atomicAdd(row + u + 2, a.z); atomicAdd(row + u + 3, a.w); }
        }
        return;
    }
    for (int f = threadIdx.x; f < per_slot; f += 256) {  // f = (dim slice * 64 + lane) * nc + component
        const int e = f / nc, c = f - e * nc;
        const int u = e * R.kpt + (c < R.kpt ? c : c - R.kpt);
        if (u >= NU) continue;
        float a = 0.f;
        if (R.row_groups <= 128) {  // eight row groups' partials at a time, the "used" bits from the ballots above
            for (int rg0 = 0; rg0 < R.row_groups; rg0 += 8) {
                float v[8];
                bool on[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int rg = rg0 + k;
                    on[k] = rg < R.row_groups && (((rg < 64 ? used_rg : used_rg2) >> (rg & 63)) & 1ull) != 0ull;
                    v[k] = R.dXp[(((size_t)(on[k] ? rg : 0) * R.npb + pb) * cap + sidx) * per_slot + f];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) a += on[k] ? v[k] : 0.f;
            }
        } else {
            for (int rg = 0; rg < R.row_groups; ++rg) {
                if (!((R.xused[((size_t)rg * R.npb + pb) * 8 + h] >> l) & 1ull)) continue;
                a += R.dXp[(((size_t)rg * R.npb + pb) * cap + sidx) * per_slot + f];
            }
        }
        add(row + (c < R.kpt ? u : R.d + u), a);
    }
}

template <int UNUSED = 0>  // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(256) void pool_dx_reduce_kernel(DxReduce R) { pool_dx_reduce_block(R, (int)blockIdx.x); }

inline DxReduce make_dx_reduce(const PoolArgs &A, int kpt, bool cplx, int row_groups) {
    DxReduce R{};
    R.dXp = A.dXp; R.xused = A.xused; R.pool = A.pool; R.g_ent = A.g_ent; R.De = A.De; R.P = A.P; R.d = A.d;
    R.npb = A.q_slices; R.halves = A.pb_halves; R.dim_slices = A.dim_slices; R.row_groups = row_groups; R.kpt = kpt;
    R.cplx = cplx ? 1 : 0; R.blocks = A.q_slices * A.pb_halves * 64;
    return R;
}

// ------------------------------------------------------------------------------------------------ launch helpers
struct PoolLaunch {
    int kpt, nw;          // units per lane, waves per workgroup (backward kernels)
    int fkpt, fnw;        // the same for the forward kernel (it amortises its wave reduction over more units per lane)
    int fwd_slices, q_slices, x_slices;
    int mfma;             // bilinear models: dense fp32 MFMA GEMMs instead of the tile kernels
    int bwd1;             // single-pass backward (pool_bwd1_kernel): q_slices = position blocks, plus the five below
    int bkpt;             // its units per lane (1, 2, or 4 for real-valued models with long rows)
    int dim_slices, pb_halves, tiles_per_wave, row_groups, cplx;
    int dense_lanes;      // lanes of every half that hold dense-prefix positions (pool_bwd1_kernel's dense pass; 0 = none)
    int tile, tile_kd, tile_ks, tile_fringe_slices;  // forward: dense prefix [0, tile_kd) on the register tile (score_pool_tile.h)
    int small, skpt, schunks;  // backward of a small problem: pool_bwd_wave_kernel (q_slices position slices, skpt units per lane, schunks dim chunks)
    int rel_copies;       // > 1: copies of the relation gradient the row backward spreads its atomics over (few relations)
    int64_t rel_elems;    // n_relation * relation_dim
    int64_t n_entity;
};

// Per-model entry points (defined in score_pool_<model>.hip): launch one of the three kernels for (head, config).
typedef int (*pool_launch_fn)(int which /*0 fwd, 1 bwd (dq + dx in one grid), 2 dx pass alone, 3 dq pass alone, 4 single-pass bwd, 5 fwd: tile + fringe, 6 bwd of a small problem (one wave per piece)*/, bool head, const PoolLaunch &L, const PoolArgs &A,
                              hipStream_t st);
int pool_launch_transe(int, bool, const PoolLaunch &, const PoolArgs &, hipStream_t);
int pool_launch_rotate(int, bool, const PoolLaunch &, const PoolArgs &, hipStream_t);
int pool_launch_complex(int, bool, const PoolLaunch &, const PoolArgs &, hipStream_t);
int pool_launch_distmult(int, bool, const PoolLaunch &, const PoolArgs &, hipStream_t);
int pool_launch_protate(int, bool, const PoolLaunch &, const PoolArgs &, hipStream_t);

template <int MODEL, bool HEAD, int KPT, int NW>
static int launch_cfg(int which, const PoolLaunch &L, const PoolArgs &A, hipStream_t st) {
    const dim3 block(NW * 64);
    if (which == 0) {
        dim3 grid((unsigned)((A.B + TI - 1) / TI), (unsigned)L.fwd_slices);
        hipLaunchKernelGGL((pool_fwd_kernel<MODEL, HEAD, KPT, NW>), grid, block, (size_t)3 * ((A.P + L.fwd_slices - 1) / L.fwd_slices) * 4, st, A);
    } else {
        // which: 1 = both passes in one grid, 2 = dx pass only, 3 = dq pass only
        const size_t rows_per = (size_t)((A.B + L.x_slices - 1) / L.x_slices);
        const size_t lds_x = ((TI + 2) * rows_per + 16) * 4;
        const size_t lds_q = ((size_t)(TI + 2) * ((A.P + L.q_slices - 1) / L.q_slices) + 32) * 4;
        const unsigned xb = which == 3 ? 0u : (unsigned)(((A.P + TI - 1) / TI) * L.x_slices);
        const unsigned qb = which == 2 ? 0u : (unsigned)(((A.B + TI - 1) / TI) * L.q_slices);
        PoolArgs A2 = A;
        A2.x_blocks = (int)xb;
        // dq workgroups ahead of the dx pass: a little under one per CU measured best (headline: 0 -> 169 us, 96..224 ->
        // 160-165 us, 256 -> 170 us; the other shapes are flat within 2 %)
        A2.q_first = xb ? (int)(qb < 160u ? qb : 160u) : 0;
        if (const char *e = getenv("MKB_POOL_QFIRST")) A2.q_first = xb ? (atoi(e) < (int)qb ? atoi(e) : (int)qb) : 0;
        hipLaunchKernelGGL((pool_bwd_kernel<MODEL, HEAD, KPT, NW>), dim3(xb + qb), block, lds_x > lds_q ? lds_x : lds_q, st, A2);
    }
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD, int KPT>
static int launch_bwd1(const PoolLaunch &L, const PoolArgs &A, hipStream_t st) {
    constexpr int NC = KPT * (ModelTraits<MODEL>::cplx_pair ? 2 : 1);
    const bool dense_cfg = A.g_blocked && L.dense_lanes > 0;
    const size_t lds_pass = (size_t)2 * L.pb_halves * L.dense_lanes * NC * 64 * 4;  // accumulator + row images of the dense slots
    const size_t lds_rows = L.tiles_per_wave == 1 ? (size_t)kBwd1Waves * TI * NC * 64 * 4 : 0;  // ... then the workgroup's query rows
    const size_t lds_main = lds_pass > lds_rows ? lds_pass : lds_rows;
    const size_t lds = dense_cfg ? lds_main + 128 + (size_t)L.pb_halves * 64 * 8  // + the block's pool-id table
                                 : (size_t)L.pb_halves * 64 * NC * 64 * 4 + 128;
    PoolArgs A2 = A;
    A2.q_slices = L.q_slices; A2.dim_slices = L.dim_slices; A2.pb_halves = L.pb_halves; A2.tiles_per_wave = L.tiles_per_wave;
    A2.dense_lanes = A.g_blocked ? L.dense_lanes : 0;  // (the dense pass reads the blocked seed layout)
    A2.lds_ids_off = (int)(lds_main / 4);
    // the dense form is compiled for the complex-modulus pair function and for TransE (pick_config never asks for it elsewhere:
    // DistMult / ComplEx take the matrix route, pRotatE's term is two transcendental chains; every instantiation is ~60 KB)
    constexpr bool kHasDense = ModelTraits<MODEL>::cplx_pair || MODEL == MKB_TRANSE;
    const bool dense = kHasDense && A2.dense_lanes > 0;
    if (!kHasDense && A2.dense_lanes > 0) return set_error(MKB_ERR_INVALID, "the dense pass is built for the complex-modulus models and TransE only");
    static LdsOptIn lds_ok[2];  // per instantiation: opt in to more than 64 KB of dynamic LDS once per device
    if (lds > 64 * 1024) {
        const void *fn = reinterpret_cast<const void *>(&pool_bwd1_kernel<MODEL, HEAD, KPT, false>);
        if constexpr (kHasDense)
            if (dense) fn = reinterpret_cast<const void *>(&pool_bwd1_kernel<MODEL, HEAD, KPT, true>);
        if (int rc = lds_ok[dense].ensure(fn, 160 * 1024)) return rc;
    }
    const int row_tiles = (A.B + TI - 1) / TI, per_group = kBwd1Waves * L.tiles_per_wave;
    const unsigned groups = (unsigned)((row_tiles + per_group - 1) / per_group);
    bool launched = false;
    if constexpr (kHasDense)
        if (dense) {
            hipLaunchKernelGGL((pool_bwd1_kernel<MODEL, HEAD, KPT, true>), dim3(groups * L.q_slices * L.dim_slices),
                               dim3(kBwd1Waves * 64), lds, st, A2);
            launched = true;
        }
    if (!launched)
        hipLaunchKernelGGL((pool_bwd1_kernel<MODEL, HEAD, KPT, false>), dim3(groups * L.q_slices * L.dim_slices),
                           dim3(kBwd1Waves * 64), lds, st, A2);
    const DxReduce R = make_dx_reduce(A2, KPT, ModelTraits<MODEL>::cplx_pair, (int)groups);
    if (A.dx_reduce_out) *A.dx_reduce_out = R;  // the caller's next launch (row backward) carries the reduction
    else hipLaunchKernelGGL(pool_dx_reduce_kernel<0>, dim3((unsigned)R.blocks), dim3(256), 0, st, R);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int launch_fwd_tile(const PoolLaunch &L, const PoolArgs &A, hipStream_t st, float *part, GemmTail *tail);  // score_pool_tile.h

template <int MODEL, bool HEAD, int KPT>
static int launch_wave(const PoolLaunch &L, const PoolArgs &A, hipStream_t st) {
    PoolArgs A2 = A;
    A2.q_slices = L.q_slices;
    const dim3 grid((unsigned)((A.B + TI - 1) / TI), (unsigned)L.q_slices, (unsigned)L.schunks);
    hipLaunchKernelGGL((pool_bwd_wave_kernel<MODEL, HEAD, KPT>), grid, dim3(64), 0, st, A2);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int launch_head(int which, const PoolLaunch &L0, const PoolArgs &A, hipStream_t st) {
    if (which == 5) return launch_fwd_tile<MODEL, HEAD>(L0, A, st, A.tile_part, A.tile_tail);
    if (which == 6) return L0.skpt == 2 ? launch_wave<MODEL, HEAD, 2>(L0, A, st) : launch_wave<MODEL, HEAD, 1>(L0, A, st);
    if (which == 4) {
        if constexpr (!ModelTraits<MODEL>::cplx_pair && MODEL != MKB_PROTATE)
            if (L0.bkpt == 4) return launch_bwd1<MODEL, HEAD, 4>(L0, A, st);
        return L0.bkpt >= 2 ? launch_bwd1<MODEL, HEAD, 2>(L0, A, st) : launch_bwd1<MODEL, HEAD, 1>(L0, A, st);
    }
    PoolLaunch L = L0;
    if (which == 0) { L.kpt = L0.fkpt; L.nw = L0.fnw; }
    if (L.kpt == 1 && L.nw == 1) return launch_cfg<MODEL, HEAD, 1, 1>(which, L, A, st);
    if (L.kpt == 2 && L.nw == 1) return launch_cfg<MODEL, HEAD, 2, 1>(which, L, A, st);
    if (L.kpt == 1 && L.nw == 2) return launch_cfg<MODEL, HEAD, 1, 2>(which, L, A, st);
    if (L.kpt == 1 && L.nw == 4) return launch_cfg<MODEL, HEAD, 1, 4>(which, L, A, st);
    if (L.kpt == 1 && L.nw == 16) return launch_cfg<MODEL, HEAD, 1, 16>(which, L, A, st);
    if (L.kpt == 2 && L.nw == 2) return launch_cfg<MODEL, HEAD, 2, 2>(which, L, A, st);
    if (L.kpt == 2 && L.nw == 4) return launch_cfg<MODEL, HEAD, 2, 4>(which, L, A, st);
    if (L.kpt == 2 && L.nw == 8) return launch_cfg<MODEL, HEAD, 2, 8>(which, L, A, st);
    if (L.kpt == 2 && L.nw == 16) return launch_cfg<MODEL, HEAD, 2, 16>(which, L, A, st);
    // (pRotatE's pair term carries a sin / cos and two divisions: its 4-units-per-lane bodies are 230 KB of code each and
    // were a seventh of the library; pick_config keeps that model at <= 2 units per lane)
    if constexpr (MODEL != MKB_PROTATE) {
        if (L.kpt == 4 && L.nw == 4) return launch_cfg<MODEL, HEAD, 4, 4>(which, L, A, st);
        if (L.kpt == 4 && L.nw == 16) return launch_cfg<MODEL, HEAD, 4, 16>(which, L, A, st);
    }
    return set_error(MKB_ERR_UNSUPPORTED, "no pooled kernel configuration (kpt=%d, nw=%d)", L.kpt, L.nw);
}

#define MKB_DEFINE_POOL_LAUNCH(fn, MODEL)                                                                       \
    int fn(int which, bool head, const PoolLaunch &L, const PoolArgs &A, hipStream_t st) {                      \
        return head ? launch_head<MODEL, true>(which, L, A, st) : launch_head<MODEL, false>(which, L, A, st); \
    }

}  // namespace mkb

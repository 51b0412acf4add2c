// pool_fwd_tile: forward of the pooled block for the complex-modulus pair function (RotatE, rotate.py:83-97) with the DENSE
// prefix of the pool on an outer-product register tile and the sparse fringe on pool_fwd_body, in ONE launch.
//
// Every row takes its first K surviving candidates of the 2K-candidate pool (negative_sampling.py:176-199), so positions
// p < K are used by (nearly) every row -- 95 % of the (row, position) pairs of an FB15k-237 batch -- while positions >= K
// are used by the rows whose filter removed something: few pairs, spread thinly up to position ~480.
//   dense prefix [0, Kd), Kd = K rounded down to the 64-position tile: workgroup = 64 rows x 64 positions x a slice of the
//     dims; a lane owns a 4 x 4 block of (row, position) sums (rows lr + 8a, positions lp + 8b: the 8 lanes of one LDS
//     instruction read 128 contiguous bytes) and walks the dims two at a time; operands come from LDS in the packed-op
//     layout (re_k, re_k+1, im_k, im_k+1), staged 16 dims at a time through registers (double buffered, one barrier per
//     chunk).  No cross-lane reduction, no per-position bookkeeping: 5 packed ops + 2 v_sqrt per two pair terms and nothing
//     else in the loop.  Pairs a row does not use are computed and ignored (cnt masks them downstream).  The dims are
//     split over workgroups to fill the chip; the partial sums [split][B][Kd] are added up by the consumer (the loss rows
//     inside mkb_pool_step: GemmTail kind 3; tile_scores_reduce_kernel elsewhere).
//   fringe [Kd, 2K): pool_fwd_body (row tiles of 8 x position slices, lanes own dims, per-row skip) as the FIRST workgroups
//     of the same grid: latency-bound, they finish in the shadow of the tile workgroups.
// Micro-benchmark of the tile alone (tools/ubench/pair_tile.hip, profiles/r03_instruction_rates_and_tile_ubench.txt):
// 51.5 us for the 1024 x 256 x 1000 block against 35.7 us of pure issue time at the measured instruction rates; the old
// kernel took 69.5 us for prefix + fringe.
#pragma once
#include "score_pool_kernels.h"

namespace mkb {

#ifndef MKB_TILE_KC
#define MKB_TILE_KC 16
#endif
#ifndef MKB_TILE_CP_PIPE
#define MKB_TILE_CP_PIPE 1  // the complex-modulus loop requests the next group's operands before it evaluates the current one, like the real-valued one (round 6 A/B, two repetitions in one call: 56.1 / 55.8 -> 55.2 / 55.3 us under the bench events; 0 = the plain loop)
#endif
constexpr int kTileKC = MKB_TILE_KC;            // dims per LDS chunk (a multiple of 16: every staging thread moves kTileKC / 16 quads of dims)
constexpr int kTileNQ = kTileKC / 16;
// (the k-pair loop below stays rolled: unrolled 2 or 4 times it is no faster -- 57.7 vs 57.0 us in round 4's A/B --, 8 times 83 us)
constexpr int kTileRows = 64, kTilePos = 64;    // workgroup tile (2 x 2 waves of 32 x 32)
constexpr int kTilePitch = kTileRows * 4 + 4;   // floats per k-pair row of the LDS image (+ 16 B pad)

struct TileArgs {
    float *part;        // [ks][B][Kd] partial sums of the dense prefix
    int Kd, ks;         // dense positions (multiple of 64), dim splits
    int fringe_tiles, fringe_slices;  // the first fringe_tiles * fringe_slices workgroups run pool_fwd_body over [Kd, P)
    int row_tiles, pos_tiles;         // dense grid: row_tiles x pos_tiles x ks workgroups behind them
};

// Round 5: the same tile for TransE's pair function |q - x| (transe.py:70-75).  A GROUP is the float4 a lane's packed step works
// on: two complex dims (re_k, re_k+1, im_k, im_k+1) for the complex modulus, four consecutive floats for the real-valued
// term; a chunk = kTileKC / 2 groups either way, so the staging, the LDS images and the 4 x 4 accumulators are shared and only
// the loads' addresses and the pair function differ.  The real-valued term is 2 VALU operations per element against 8
// ds_read_b128 per 64 elements and lane: the LDS array is busy about half the time.
template <int MODEL, bool HEAD>
__device__ __forceinline__ f2 tile_pair_term(float4 q, float4 x) {
    if constexpr (ModelTraits<MODEL>::cplx_pair) {
        return pair_term_cmod2(f2{q.x, q.y}, f2{q.z, q.w}, f2{x.x, x.y}, f2{x.z, x.w});
    } else {
        static_assert(MODEL == MKB_TRANSE, "real-valued tile: TransE's |q - x| only");
        const f2 a = HEAD ? f2{x.x, x.y} + f2{q.x, q.y} : f2{q.x, q.y} - f2{x.x, x.y};
        const f2 b = HEAD ? f2{x.z, x.w} + f2{q.z, q.w} : f2{q.z, q.w} - f2{x.z, x.w};
        // |a| + |b| as ONE v_add_f32 with both source modifiers (left to itself the compiler clears the sign bits with four
        // v_and_b32 and adds packed: 8 instructions per group of four elements instead of 5)
        f2 r;
        asm("v_add_f32 %0, |%1|, |%2|" : "=v"(r.x) : "v"(a.x), "v"(b.x));
        asm("v_add_f32 %0, |%1|, |%2|" : "=v"(r.y) : "v"(a.y), "v"(b.y));
        return r;
    }
}

template <int MODEL, bool HEAD, int KPT>
__global__ __launch_bounds__(256) void pool_fwd_tile_kernel(PoolArgs A, TileArgs T) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    extern __shared__ __attribute__((aligned(16))) int lds_tile_dyn[];  // the fringe workgroups' position lists
    const int b = (int)blockIdx.x;
    const int n_fringe = T.fringe_tiles * T.fringe_slices;
    if (b < n_fringe) {  // (workgroup-uniform)
        pool_fwd_body<MODEL, HEAD, KPT, 4>(A, b % T.fringe_tiles, T.fringe_tiles, b / T.fringe_tiles, T.fringe_slices, lds_tile_dyn);
        return;
    }
    __shared__ __attribute__((aligned(16))) float sq[2][kTileKC / 2][kTilePitch];
    __shared__ __attribute__((aligned(16))) float sx[2][kTileKC / 2][kTilePitch];
    int t = b - n_fringe;
    const int z = t % T.ks; t /= T.ks;
    const int i0 = (t % T.row_tiles) * kTileRows, p0 = (t / T.row_tiles) * kTilePos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 7, lp = lane >> 3;
    const int wr = (wave >> 1) * 32, wp = (wave & 1) * 32;
    // k counts "dims": complex dims for the complex modulus (two per group), PAIRS of floats for the real-valued term (two per
    // group as well: a chunk of kTileKC dims = kTileKC / 2 groups = 32 floats of a row either way)
    const int d = CP ? A.d : (int)(A.De / 2);
    const int kper = ((d + T.ks - 1) / T.ks + kTileKC - 1) / kTileKC * kTileKC;
    const int k_lo = z * kper, k_hi = min(d, k_lo + kper);

    // staging: thread -> (row of the tile, quad of dims): two groups per operand and chunk (complex: one float4 of each half,
    // real: two consecutive float4)
    const int srow = tid >> 2, skq = tid & 3;
    const float *qsrc = A.Q + (int64_t)min(i0 + srow, A.B - 1) * A.De;
    const float *xsrc = A.ent + A.pool[min(p0 + srow, T.Kd - 1)] * A.De;
    float4 rq_re[kTileNQ], rq_im[kTileNQ], rx_re[kTileNQ], rx_im[kTileNQ];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < kTileNQ; ++j) {
            const int kq = k0 + 4 * (skq + 4 * j);  // first dim of this thread's j-th quad
            if constexpr (CP) {
                const int k = min(kq, d - 4);
                rq_re[j] = *reinterpret_cast<const float4 *>(qsrc + k);
                rq_im[j] = *reinterpret_cast<const float4 *>(qsrc + d + k);
                rx_re[j] = *reinterpret_cast<const float4 *>(xsrc + k);
                rx_im[j] = *reinterpret_cast<const float4 *>(xsrc + d + k);
            } else {  // groups g, g + 1 = floats [2 k, 2 k + 8) of the row (De is a multiple of 4: the last group may stand alone)
                const int f0 = min(2 * kq, (int)A.De - 4), f1 = min(2 * kq + 4, (int)A.De - 4);
                rq_re[j] = *reinterpret_cast<const float4 *>(qsrc + f0);
                rq_im[j] = *reinterpret_cast<const float4 *>(qsrc + f1);
                rx_re[j] = *reinterpret_cast<const float4 *>(xsrc + f0);
                rx_im[j] = *reinterpret_cast<const float4 *>(xsrc + f1);
            }
        }
    };
    auto lstore = [&](int buf, int k0) {
        const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < kTileNQ; ++j) {
            const int kq = k0 + 4 * (skq + 4 * j), g0 = 2 * (skq + 4 * j);
            if constexpr (CP) {
                const bool ok = kq < k_hi;  // dims past the split's range contribute |0 - 0| = 0
                const float4 a = ok ? rq_re[j] : zz, bq = ok ? rq_im[j] : zz, c = ok ? rx_re[j] : zz, e = ok ? rx_im[j] : zz;
                *reinterpret_cast<float4 *>(&sq[buf][g0][srow * 4]) = make_float4(a.x, a.y, bq.x, bq.y);
                *reinterpret_cast<float4 *>(&sq[buf][g0 + 1][srow * 4]) = make_float4(a.z, a.w, bq.z, bq.w);
                *reinterpret_cast<float4 *>(&sx[buf][g0][srow * 4]) = make_float4(c.x, c.y, e.x, e.y);
                *reinterpret_cast<float4 *>(&sx[buf][g0 + 1][srow * 4]) = make_float4(c.z, c.w, e.z, e.w);
            } else {
                const bool ok0 = kq < k_hi, ok1 = kq + 2 < k_hi;
                *reinterpret_cast<float4 *>(&sq[buf][g0][srow * 4]) = ok0 ? rq_re[j] : zz;
                *reinterpret_cast<float4 *>(&sq[buf][g0 + 1][srow * 4]) = ok1 ? rq_im[j] : zz;
                *reinterpret_cast<float4 *>(&sx[buf][g0][srow * 4]) = ok0 ? rx_re[j] : zz;
                *reinterpret_cast<float4 *>(&sx[buf][g0 + 1][srow * 4]) = ok1 ? rx_im[j] : zz;
            }
        }
    };
    f2 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f2{0.f, 0.f};
    if (k_lo < k_hi) {
        gload(k_lo);
        lstore(0, k_lo);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += kTileKC) {
        const bool more = k0 + kTileKC < k_hi;
        if (more) gload(k0 + kTileKC);  // the next chunk's global loads fly under this chunk's pair math
        if constexpr (CP && MKB_TILE_CP_PIPE == 0) {
#pragma unroll 1
            for (int kp = 0; kp < kTileKC / 2; ++kp) {
                float4 q[4], x[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) q[a] = *reinterpret_cast<const float4 *>(&sq[buf][kp][(wr + lr + 8 * a) * 4]);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = *reinterpret_cast<const float4 *>(&sx[buf][kp][(wp + lp + 8 * c) * 4]);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[a][c] += tile_pair_term<MODEL, HEAD>(q[a], x[c]);
            }
        } else {
            // the real-valued term is ~100 VALU operations per group against the round trip of its eight LDS reads, with two
            // waves per SIMD to hide it: the next group's operands are requested before this group's terms are evaluated
            // (two register sets that take turns: written as "q = qn; qn = load(kp + 1)" the compiler folds the copy away
            // and waits for the loads it has just issued)
            float4 qa[4], xa[4], qb[4], xb[4];
            auto lds_group = [&](int kp, float4 (&q)[4], float4 (&x)[4]) {
#pragma unroll
                for (int a = 0; a < 4; ++a) q[a] = *reinterpret_cast<const float4 *>(&sq[buf][kp][(wr + lr + 8 * a) * 4]);
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = *reinterpret_cast<const float4 *>(&sx[buf][kp][(wp + lp + 8 * c) * 4]);
            };
            auto terms = [&](const float4 (&q)[4], const float4 (&x)[4]) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] += tile_pair_term<MODEL, HEAD>(q[a], x[c]);
            };
            lds_group(0, qa, xa);
#pragma unroll 1
            for (int kp = 0; kp < kTileKC / 2; kp += 2) {
                lds_group(kp + 1, qb, xb);
                terms(qa, xa);
                lds_group(min(kp + 2, kTileKC / 2 - 1), qa, xa);
                terms(qb, xb);
            }
        }
        if (more) lstore(buf ^ 1, k0 + kTileKC);
        __syncthreads();
        buf ^= 1;
    }
    float *out = T.part + (int64_t)z * A.B * T.Kd;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + wr + lr + 8 * a;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p = p0 + wp + lp + 8 * c;
            if (i < A.B && p < T.Kd) out[(int64_t)i * T.Kd + p] = acc[a][c].x + acc[a][c].y;
        }
    }
}

// S[i][p] = c0 + c1 * sum_z part[z][i][p] for p < Kd where row i uses p, 0 where it does not (fixed order: deterministic).
// The consumer inside mkb_pool_step is the loss kernel (GemmTail kind 3); this kernel serves the calls that hand S out.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void tile_scores_reduce_kernel(const float *__restrict__ part, const uint16_t *__restrict__ cnt,
                                                                 float *__restrict__ S, int B, int P, int Kd, int ks, float c0, float c1) {
    const int64_t n = (int64_t)B * Kd;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / Kd, p = e - i * Kd;
        float s = 0.f;
        for (int z0 = 0; z0 < ks; z0 += 8) {  // eight splits' loads at a time, added in split order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(int64_t)min(z0 + k, ks - 1) * n + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += z0 + k < ks ? v[k] : 0.f;
        }
        S[i * P + p] = cnt[i * P + p] ? c0 + c1 * s : 0.f;
    }
}

// tail: non-null = the consumer adds the partial sums up itself (described in *tail, kind 3); null = S is finished here
template <int MODEL, bool HEAD>
static int launch_fwd_tile(const PoolLaunch &L, const PoolArgs &A0, hipStream_t st, float *part, GemmTail *tail) {
    if constexpr (!ModelTraits<MODEL>::cplx_pair && MODEL != MKB_TRANSE) {
        return set_error(MKB_ERR_UNSUPPORTED, "pool_fwd_tile: complex-modulus pair function and TransE only");
    } else {
        PoolArgs A = A0;
        TileArgs T{};
        T.part = part; T.Kd = L.tile_kd; T.ks = L.tile_ks;
        T.row_tiles = (A.B + kTileRows - 1) / kTileRows; T.pos_tiles = T.Kd / kTilePos;
        A.p_lo = T.Kd;
        const bool fringe = A.P > T.Kd;
        T.fringe_tiles = fringe ? (A.B + TI - 1) / TI : 0;
        T.fringe_slices = fringe ? L.tile_fringe_slices : 0;
        // measurement only, WRONG scores ('d' = the dense tiles alone, 'f' = the fringe alone): honoured only together with
        // MKB_MEASURE_WRONG_RESULTS=1, so that a stray variable cannot silently break a run
        if (const char *e = getenv("MKB_MEASURE_WRONG_RESULTS") ? getenv("MKB_POOL_TILE_ONLY") : nullptr) {
            if (e[0] == 'd') { T.fringe_tiles = 0; T.fringe_slices = 0; }
            if (e[0] == 'f') T.pos_tiles = 0;
        }
        const size_t lds = (T.fringe_tiles > 0) ? (size_t)3 * ((A.P - T.Kd + T.fringe_slices - 1) / T.fringe_slices) * 4 : 0;
        const unsigned blocks = (unsigned)(T.fringe_tiles * T.fringe_slices + T.row_tiles * T.pos_tiles * T.ks);
        if (L.fkpt == 4) hipLaunchKernelGGL((pool_fwd_tile_kernel<MODEL, HEAD, 4>), dim3(blocks), dim3(256), lds, st, A, T);
        else hipLaunchKernelGGL((pool_fwd_tile_kernel<MODEL, HEAD, 2>), dim3(blocks), dim3(256), lds, st, A, T);
        if (tail) {
            *tail = GemmTail{3, part, A.S, nullptr, A.B, T.Kd, T.ks, A.P, (int64_t)A.B * T.Kd, A.c0, A.c1, nullptr, 0};
        } else {
            hipLaunchKernelGGL(tile_scores_reduce_kernel<0>, dim3(512), dim3(256), 0, st, part, A.cnt, A.S, A.B, A.P, T.Kd, T.ks, A.c0, A.c1);
        }
        MKB_LAUNCH_CHECK();
        return MKB_OK;
    }
}

}  // namespace mkb

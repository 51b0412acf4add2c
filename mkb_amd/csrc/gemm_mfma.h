// fp32 MFMA GEMM used by the pooled path of the bilinear models (ComplEx, DistMult).
//
// For those models the pair function is a plain dot product, so with mkb's shared candidate pool the negative block
// is three dense GEMMs:
//     S  [B, P]  = Q [B, De] . Xp[P, De]^T          Xp = ent[pool]   (gathered through the row index, never copied)
//     dQ [B, De] = G [B, P]  . Xp[P, De]
//     dXp[P, De] = G^T[P, B] . Q [B, De]            accumulated into g_ent[pool[p]] (atomics: pool ids may repeat)
// G is zero where a row does not use a position, so no masking is needed.  gfx950 has no TF32/xf32; the exact
// fp32-input matrix instruction v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate with bitwise fmaf-chain
// numerics, which is what the 1e-4 parity budget wants.
//
// One kernel, 256 lanes = 4 waves, 64 x 64 output tile (one 32 x 32 MFMA tile per wave), K stepped in chunks of
// 32 through LDS.  Operand layouts are template flags; tiles are always read from global memory with the lanes
// along the CONTIGUOUS index (coalesced 128-B segments) and stored in LDS as [m][k] / [n][k] with a +1 pad so the
// per-MFMA fragment reads (lane l: row l&31, k = kk + (l>>5)) are bank-conflict free.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace mkb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { GEMM_STORE = 0, GEMM_STORE_AFFINE = 1, GEMM_ATOMIC_ROWS = 2 };
constexpr bool kGemmBf16x3Default = true;  // MKB_GEMM_BF16X3=0: the fp32-input matrix instruction instead (gemm128_f32_mfma_kernel)

struct GemmArgs {
    const float *A, *B;
    float *C;
    const int64_t *b_idx;   // optional row indirection of B (B_NK: indexed by n; B_KN: indexed by k)
    int64_t b_rows;         // ... rows of the table b_idx points into (0 = unknown: the bf16 kernel's 32-bit offsets are not used)
    const int64_t *c_idx;   // GEMM_ATOMIC_ROWS: output row m goes to C[c_idx[m]]
    const float *scale_dev; // optional device scalar multiplied into c1 (pRotatE-style), unused for the dot models
    int M, N, K;
    int ksplit;             // K is split over gridDim.z; STORE epilogues then write partial z at C + z * M * ldc
    int64_t lda, ldb, ldc;
    float c0, c1;           // GEMM_STORE_AFFINE: C = c0 + c1 * acc
    // Used pool depth (optional).  depth[i] = one past the last pool position batch row i uses; positions beyond the deepest
    // one any relevant row uses carry no information (scores nobody reads, gradient seeds that are exactly 0):
    //   depth_mode 1: N runs over pool positions, M over batch rows: column tiles beyond the row tile's depth are not computed
    //   depth_mode 2: K runs over pool positions, M over batch rows: the K range is cut at the row tile's depth
    //   depth_mode 3: M runs over pool positions: row tiles beyond the depth of ALL n_depth batch rows are not computed
    const int *depth;
    int depth_mode, n_depth;
};

// -> (skip this workgroup, k limit)
__device__ __forceinline__ bool gemm_depth_cut(const GemmArgs &G, int m0, int tm, int n0, int *s_red, int &k_cut) {
    k_cut = 0x7fffffff;
    if (!G.depth) return false;
    if (G.depth_mode == 3) return m0 >= block_max_i32(G.depth, 0, G.n_depth, s_red);
    const int lim = block_max_i32(G.depth, m0, min(m0 + tm, G.M), s_red);
    if (G.depth_mode == 1) return n0 >= lim;
    k_cut = (lim + 3) & ~3;
    return false;
}

// A_MK: A(m,k) = A[m*lda + k]   (k contiguous)      else A_KM: A(m,k) = A[k*lda + m]   (m contiguous)
// B_NK: B(k,n) = B[idx(n)*ldb + k] (k contiguous)   else B_KN: B(k,n) = B[idx(k)*ldb + n] (n contiguous)
// TM = 64 (4 waves) or 32 (2 waves, twice the workgroups: for the short-and-wide forward product).
template <bool A_MK, bool B_NK, int EPI, int TM>
__global__ __launch_bounds__(TM * 4) void gemm_f32_mfma_kernel(GemmArgs G) {
    constexpr int TN = 64, KC = 32, LD = KC + 1, T = TM * 4;
    constexpr int NA = TM * KC / T, NB = TN * KC / T;  // elements per lane and K chunk
    __shared__ float sA[TM * LD];
    __shared__ float sB[TN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ra[NA], rb[NB];
    const int kper_ = ((G.K + gridDim.z - 1) / gridDim.z + KC - 1) / KC * KC;  // K range of this split, chunk aligned
    __shared__ int s_red[16];
    int k_cut;
    if (gemm_depth_cut(G, m0, TM, n0, s_red, k_cut)) return;  // (workgroup-uniform)
    const int k_lo = blockIdx.z * kper_, k_hi = min(min(G.K, k_cut), k_lo + kper_);

    // lanes run along the contiguous index of each operand (coalesced); e = element number within the lane's share
    auto a_coord = [&](int e, int &r, int &kk) {
        if constexpr (A_MK) { kk = tid & 31; r = (tid >> 5) + e * (T / 32); }
        else { r = tid % TM; kk = tid / TM + e * (T / TM); }
    };
    auto b_coord = [&](int e, int &r, int &kk) {
        if constexpr (B_NK) { kk = tid & 31; r = (tid >> 5) + e * (T / 32); }
        else { r = tid & 63; kk = (tid >> 6) + e * (T / 64); }
    };
    // B_NK: the rows a lane stages are the same for every K chunk: resolve the (gathered) row pointers once
    const float *brow[NB];
    bool brow_ok[NB];
    if constexpr (B_NK) {
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            int r, kk;
            b_coord(e, r, kk);
            const int n = n0 + r;
            brow_ok[e] = n < G.N;
            const int64_t sel = brow_ok[e] ? n : 0;
            brow[e] = G.B + (G.b_idx ? G.b_idx[sel] : sel) * G.ldb;
        }
    }
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            int r, kk;
            a_coord(e, r, kk);
            const int m = m0 + r, k = k0 + kk;
            const bool ok = m < G.M && k < k_hi;
            const int64_t off = A_MK ? (int64_t)(ok ? m : 0) * G.lda + (ok ? k : 0) : (int64_t)(ok ? k : 0) * G.lda + (ok ? m : 0);
            const float v = G.A[off];  // unconditional load from a clamped address, then select
            ra[e] = ok ? v : 0.f;
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            int r, kk;
            b_coord(e, r, kk);
            const int n = n0 + r, k = k0 + kk;
            if constexpr (B_NK) {
                const bool ok = brow_ok[e] && k < k_hi;
                const float v = brow[e][ok ? k : 0];
                rb[e] = ok ? v : 0.f;
            } else {
                const bool ok = n < G.N && k < k_hi;
                const int64_t sel = ok ? k : 0;
                const int64_t row = G.b_idx ? G.b_idx[sel] : sel;
                const float v = G.B[row * G.ldb + (ok ? n : 0)];
                rb[e] = ok ? v : 0.f;
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int e = 0; e < NA; ++e) { int r, kk; a_coord(e, r, kk); sA[r * LD + kk] = ra[e]; }
#pragma unroll
        for (int e = 0; e < NB; ++e) { int r, kk; b_coord(e, r, kk); sB[r * LD + kk] = rb[e]; }
    };

    if (k_lo < k_hi) load_tiles(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        store_tiles();
        __syncthreads();
        if (k0 + KC < k_hi) load_tiles(k0 + KC);  // next chunk in flight under this chunk's MFMAs
        const float *pa = sA + (wm + (lane & 31)) * LD + (lane >> 5);
        const float *pb = sB + (wn + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pb[kk], acc, 0, 0, 0);
        __syncthreads();
    }
    // ---- epilogue: acc[reg] is C(row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31) of the wave's 32x32 tile
    const int n = n0 + wn + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m < G.M && n < G.N) {
            const float v = acc[reg];
            float *Cz = G.C + (int64_t)blockIdx.z * G.M * G.ldc;  // partial buffer of this K split (z = 0: C itself)
            if constexpr (EPI == GEMM_STORE) Cz[(int64_t)m * G.ldc + n] = v;
            else if constexpr (EPI == GEMM_STORE_AFFINE) Cz[(int64_t)m * G.ldc + n] = (gridDim.z > 1) ? v : G.c0 + G.c1 * v;
            else if (v != 0.f) atomicAdd(G.C + G.c_idx[m] * G.ldc + n, v);
        }
    }
}

// ---- second kernel: 128 x 128 workgroup tile, 2 x 2 MFMA tiles (64 x 64) per wave ---------------------------------------
// The 64 x 64 kernel above gives a wave 16 MFMAs (1024 cycles) per K chunk against a global-load + LDS round trip of several
// thousand cycles per chunk: PMC showed the matrix pipe busy 25-30 % of the launch and the waves parked in s_waitcnt /
// s_barrier 55-60 % of their lifetime.  Here a wave owns four accumulators: 64 MFMAs (4096 cycles) per chunk for twice the
// staged bytes, fragments are read once per two MFMAs, global loads are 16 bytes wide along the contiguous index, and the
// next chunk's loads are in flight under the current chunk's MFMAs.  Requires M, N, lda, ldb multiples of 4 (16-byte rows).
// TN = 128: 2 x 2 MFMA tiles per wave; TN = 64: 2 x 1 (twice the workgroups: products that would otherwise need a K split).
template <bool A_MK, bool B_NK, int EPI, int TN>
__global__ __launch_bounds__(256) void gemm128_f32_mfma_kernel(GemmArgs G) {
    constexpr int TM = 128, KC = 32, LD = KC + 1, T = 256, NT = TN / 64;
    constexpr int V = TM * KC / 4 / T, VB = TN * KC / 4 / T;  // float4 per lane and K chunk: A (= 4), B (= 4 or 2)
    extern __shared__ __attribute__((aligned(16))) float lds_g[];
    // two stages of (A tile, B tile): chunk k + 1 is written while chunk k is multiplied -- ONE barrier per chunk (round 3: a
    // single stage, i.e. a barrier after the writes and another after the multiplies)
    constexpr int STAGE = (TM + TN) * LD;
    float *sA = lds_g, *sB = lds_g + TM * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (TN / 2);
    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int kper_ = ((G.K + gridDim.z - 1) / gridDim.z + KC - 1) / KC * KC;
    __shared__ int s_red[16];
    int k_cut;
    if (gemm_depth_cut(G, m0, TM, n0, s_red, k_cut)) return;  // (workgroup-uniform) a tile nobody needs: not even written
    const int k_lo = blockIdx.z * kper_, k_hi = min(min(G.K, k_cut), k_lo + kper_);

    // element e of this lane's share: (row r, first k kk) for k-contiguous operands (float4 along k),
    //                                  (first row r, k kk) for row-contiguous operands (float4 along m / n)
    auto coord = [&](bool k_major, int rows, int e, int &r, int &kk) {
        const int f = tid + e * T;                      // float4 index within the rows x 32 chunk
        if (k_major) { r = f >> 3; kk = (f & 7) * 4; }  // 8 float4 per row of 32 k
        else { kk = f / (rows / 4); r = (f % (rows / 4)) * 4; }  // rows / 4 float4 per k column
    };
    // Loads are UNCONDITIONAL from clamped addresses (no select on the loaded value: the compiler turns such a select into a
    // branch around the load and then waits for the data before the MFMAs).  Rows beyond M / N only feed output rows /
    // columns that are never stored; k beyond the range is zeroed when the chunk is written to LDS.  Row indirection:
    // B_NK rows are fixed for the whole K loop (resolved once); B_KN rows change with k: their ids are staged in LDS first
    // (one dependent global round trip per workgroup instead of one per load).
    int *s_idx = reinterpret_cast<int *>(lds_g + 2 * STAGE);  // [kper_] row ids of the K range (B_KN with b_idx)
    int64_t brow[VB];
    if constexpr (B_NK) {
#pragma unroll
        for (int e = 0; e < VB; ++e) {
            int r, kk;
            coord(true, TN, e, r, kk);
            const int64_t sel = min(n0 + r, G.N - 1);
            brow[e] = (G.b_idx ? G.b_idx[sel] : sel) * G.ldb;
        }
    } else if (G.b_idx) {
        for (int k = k_lo + tid; k < k_hi; k += T) s_idx[k - k_lo] = (int)G.b_idx[k];
        __syncthreads();
    }
    float4 ra[V], rb[VB];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            int r, kk;
            coord(A_MK, TM, e, r, kk);
            // (the float4 runs along k for A_MK, along m otherwise: only then must it start four rows before the end.  Rounds 3-4
            // clamped to M - 4 in both cases: the LAST THREE ROWS of a k-contiguous A -- batch rows B-3 .. B-1 of the score and
            // dQ products -- were computed from row M - 4; found by comparing against the VALU route on every row)
            const int m = min(m0 + r, A_MK ? G.M - 1 : G.M - 4), k = min(k0 + kk, A_MK ? k_hi - 4 : k_hi - 1);
            ra[e] = *reinterpret_cast<const float4 *>(G.A + (A_MK ? (int64_t)m * G.lda + k : (int64_t)k * G.lda + m));
        }
#pragma unroll
        for (int e = 0; e < VB; ++e) {
            int r, kk;
            coord(B_NK, TN, e, r, kk);
            if constexpr (B_NK) {
                rb[e] = *reinterpret_cast<const float4 *>(G.B + brow[e] + min(k0 + kk, k_hi - 4));
            } else {
                const int k = min(k0 + kk, k_hi - 1), n = min(n0 + r, G.N - 4);
                const int64_t row = G.b_idx ? (int64_t)s_idx[k - k_lo] : (int64_t)k;
                rb[e] = *reinterpret_cast<const float4 *>(G.B + row * G.ldb + n);
            }
        }
    };
    // full_c: the whole chunk lies inside the K range (every chunk but possibly the last): no per-element range selects
    auto store_tiles = [&](int k0, int stage, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        float *sA = lds_g + stage * STAGE, *sB = sA + TM * LD;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            int r, kk;
            coord(A_MK, TM, e, r, kk);
            float *d = sA + r * LD + kk;
            if constexpr (A_MK) {  // float4 along k: zero the part beyond the range
                d[0] = FULL || k0 + kk < k_hi ? ra[e].x : 0.f; d[1] = FULL || k0 + kk + 1 < k_hi ? ra[e].y : 0.f;
                d[2] = FULL || k0 + kk + 2 < k_hi ? ra[e].z : 0.f; d[3] = FULL || k0 + kk + 3 < k_hi ? ra[e].w : 0.f;
            } else {
                const bool ok = FULL || k0 + kk < k_hi;
                d[0] = ok ? ra[e].x : 0.f; d[LD] = ok ? ra[e].y : 0.f; d[2 * LD] = ok ? ra[e].z : 0.f; d[3 * LD] = ok ? ra[e].w : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < VB; ++e) {
            int r, kk;
            coord(B_NK, TN, e, r, kk);
            float *d = sB + r * LD + kk;
            if constexpr (B_NK) {
                d[0] = FULL || k0 + kk < k_hi ? rb[e].x : 0.f; d[1] = FULL || k0 + kk + 1 < k_hi ? rb[e].y : 0.f;
                d[2] = FULL || k0 + kk + 2 < k_hi ? rb[e].z : 0.f; d[3] = FULL || k0 + kk + 3 < k_hi ? rb[e].w : 0.f;
            } else {
                const bool ok = FULL || k0 + kk < k_hi;
                d[0] = ok ? rb[e].x : 0.f; d[LD] = ok ? rb[e].y : 0.f; d[2 * LD] = ok ? rb[e].z : 0.f; d[3 * LD] = ok ? rb[e].w : 0.f;
            }
        }
    };

    auto store_chunk = [&](int k0, int stage) {
        if (k0 + KC <= k_hi) store_tiles(k0, stage, std::true_type{});
        else store_tiles(k0, stage, std::false_type{});
    };
    if (k_lo < k_hi) {
        load_tiles(k_lo);
        store_chunk(k_lo, 0);
    }
    __syncthreads();
    int stage = 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        const bool more = k0 + KC < k_hi;
        if (more) load_tiles(k0 + KC);  // next chunk in flight under this chunk's 64 MFMAs
        const float *pa = sA + stage * STAGE + (wm + (lane & 31)) * LD + (lane >> 5);
        const float *pb = sB + stage * STAGE + (wn + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const float a0 = pa[kk], a1 = pa[32 * LD + kk], b0 = pb[kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            if constexpr (NT == 2) {
                const float b1 = pb[32 * LD + kk];
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        // the other stage was last read during the previous chunk, and every wave has passed that chunk's barrier since
        if (more) store_chunk(k0 + KC, stage ^ 1);
        __syncthreads();
        stage ^= 1;
    }
    float *Cz = G.C + (int64_t)blockIdx.z * G.M * G.ldc;  // partial buffer of this K split (z = 0: C itself)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = n0 + wn + 32 * b + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm + 32 * a + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < G.M && n < G.N) {
                    const float v = acc[a][b][reg];
                    if constexpr (EPI == GEMM_STORE) Cz[(int64_t)m * G.ldc + n] = v;
                    else if constexpr (EPI == GEMM_STORE_AFFINE) Cz[(int64_t)m * G.ldc + n] = (gridDim.z > 1) ? v : G.c0 + G.c1 * v;
                    else if (v != 0.f) atomicAdd(G.C + G.c_idx[m] * G.ldc + n, v);
                }
            }
        }
}

// ---- third kernel: the same 128 x TN tile on the bf16 matrix pipe, fp32 operands split three ways ------------------------
// gfx950 multiplies bf16 sixteen times faster than fp32 (v_mfma_f32_32x32x16_bf16: 32 x 32 x 16 in 8 passes; the fp32-input
// v_mfma_f32_32x32x2_f32 needs 16 passes for 32 x 32 x 2).  An fp32 value x is EXACTLY hi + mid + lo with three bf16 numbers
// (8 significant bits each, rounded to nearest: x - hi and (x - hi) - mid are exact in fp32, lo carries the last rounding), so
//     a . b = (ah + am + al)(bh + bm + bl)  ~  ah bh + ah bm + am bh + ah bl + al bh + am bm
// -- the three products left out are below 2^-24 |a||b|, the size of fp32's own rounding -- every kept product of two bf16
// numbers is exact in fp32 and the sums accumulate in fp32 inside the matrix unit.  Six bf16 instructions do the work of
// eight fp32 ones at a sixteenth of the passes each: 2.7x the matrix rate for results within ~2e-7 |a||b| per term of the
// fp32 kernel's (the parity budget is 1e-4).  The operands are split when a K chunk is written to LDS (5.5 VALU operations
// per value: split3_bf16_pair), three bf16 planes per operand, rows 80 bytes apart (64 bytes of k + 16 of pad: the 16-byte fragment
// reads of 8 consecutive rows fall on 8 different bank groups).  k-contiguous operands are loaded as before (float4 along k,
// 8-byte LDS stores); row-contiguous operands are loaded one value per lane and k -- lanes along the rows, coalesced -- so
// that a lane holds 8 or 16 consecutive k of ONE row and writes them with 16-byte stores (4-byte stores of the float4
// form land 64 lanes on four banks: tried in round 4 with fp32 fragments, 29 -> 39 us).  One LDS stage (46 / 61 KB), two
// barriers per chunk: two workgroups share a CU and fill each other's staging phases.  Measured at the ComplEx shape
// (1024 x 512 x 2000 and its two transposes; fp32 kernel -> this one): score product 26.2 -> 19.7 us, dQ 21.5 -> 20.7,
// dX 28.8 -> 23.6: these products are short (8-16 chunks per workgroup), what is left is their prologues and epilogues.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kBfPitch = 80;  // bytes between rows of a bf16 plane

// two values -> one word per plane (element k in the low half, k + 1 in the high half): v_cvt_pk_bf16_f32 rounds to nearest and
// packs in one instruction; x - hi and (x - hi) - mid are exact in fp32.  11 VALU operations per pair.
// Non-finite operands: +-Inf (and |x| > ~3.39e38, which rounds to Inf in bf16) give hi = Inf, x - hi = NaN, i.e. a NaN product
// where the fp32 kernel and the reference give +-Inf; NaN stays NaN.  Either way the table is beyond repair -- no guard is
// spent on it in the staging loop (MKB_GEMM_BF16X3=0 selects the fp32-input instruction, which propagates Inf like the reference).
__device__ __forceinline__ void split3_bf16_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){s0, s1}, bf16x2));
}

// one operand's share of a K chunk: ROWS x 32 values per 256 lanes
template <bool K_MAJOR, int ROWS>
struct Bf16Stage {
    static constexpr int NEL = ROWS * 32 / 256;  // values per lane and chunk: 16 (128 rows) or 8 (64 rows)
    float v[NEL];
    // k-contiguous: NEL / 4 float4, f = tid + e * 256 -> row f >> 3, k (f & 7) * 4.  row-contiguous: row tid % ROWS, k (tid / ROWS) * NEL + j
    static constexpr int NROWS = K_MAJOR ? NEL / 4 : 1;  // rows a lane touches (k-contiguous: one per float4)
    __device__ __forceinline__ static int row_of(int tid, int e) { return K_MAJOR ? (tid + e * 256) >> 3 : tid % ROWS; }
    __device__ __forceinline__ static int k_of(int tid, int e) {  // chunk-relative k of the lane's e-th float4 / of its first value
        return K_MAJOR ? ((tid + e * 256) & 7) * 4 : __builtin_amdgcn_readfirstlane(tid / ROWS) * NEL;
    }
    // Buffer loads (one descriptor per operand, built from kernel arguments: scalar registers): the address of a load is
    // descriptor base + voff (per lane, bytes; the same in every K chunk: computed once by the caller) + soff (scalar, bytes):
    //   k-contiguous:   voff[e] = (row * ld + k of the float4) * 4,   soff = k0 * 4
    //   row-contiguous: voff[0] = row * 4,                           soff(j) = (row of k0 + kk0 + j) * ld * 4 -- whole waves
    //     share a k group (ROWS is a multiple of 64), so k is wave-uniform and this is scalar arithmetic
    // With 64-bit flat addresses the staging of the row-contiguous operands spent more VALU instructions on addresses than on
    // the split (1500 VALU instructions around 49 matrix instructions per chunk pair).
    //   (k-contiguous, last chunk of a range: a float4 whose k lies beyond the range is pulled back onto the range's last four
    //   values -- soff(e) is then per lane; its values are zeroed at the store)
    template <typename Rsrc, typename SoffFn>
    __device__ __forceinline__ void load(Rsrc rs, const int (&voff)[NROWS], int tid, SoffFn soff) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if constexpr (K_MAJOR) {
#pragma unroll
            for (int e = 0; e < NEL / 4; ++e) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[e] + soff(e), 0, 0);
                v[4 * e] = __uint_as_float(q.x); v[4 * e + 1] = __uint_as_float(q.y);
                v[4 * e + 2] = __uint_as_float(q.z); v[4 * e + 3] = __uint_as_float(q.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NEL; ++j) v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[0], soff(j), 0));
        }
    }
    // planes: [3][ROWS] rows of kBfPitch bytes; values whose k is not below k_lim (chunk-relative) are written as 0 (FULL: the
    // whole chunk lies inside the K range -- every chunk but possibly the last: no selects)
    template <bool FULL>
    __device__ __forceinline__ void store_as(int tid, unsigned char *planes, int k_lim) const {
        constexpr int PLANE = ROWS * kBfPitch;
        auto val = [&](int idx, int k) { return FULL || k < k_lim ? v[idx] : 0.f; };
        if constexpr (K_MAJOR) {
#pragma unroll
            for (int e = 0; e < NEL / 4; ++e) {
                const int f = tid + e * 256, r = f >> 3, kk = (f & 7) * 4;
                unsigned h[2], m[2], l[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split3_bf16_pair(val(4 * e + 2 * j, kk + 2 * j), val(4 * e + 2 * j + 1, kk + 2 * j + 1), h[j], m[j], l[j]);
                unsigned char *d = planes + r * kBfPitch + kk * 2;
                *reinterpret_cast<uint2 *>(d) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(m[0], m[1]);
                *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(l[0], l[1]);
            }
        } else {
            const int r = tid % ROWS, kk0 = (tid / ROWS) * NEL;
            unsigned char *d = planes + r * kBfPitch + kk0 * 2;
#pragma unroll
            for (int q = 0; q < NEL / 8; ++q) {  // eight values -> one 16-byte store per plane
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    split3_bf16_pair(val(8 * q + 2 * j, kk0 + 8 * q + 2 * j), val(8 * q + 2 * j + 1, kk0 + 8 * q + 2 * j + 1), h[j], m[j], l[j]);
                *reinterpret_cast<uint4 *>(d + 16 * q) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4 *>(d + 16 * q + PLANE) = make_uint4(m[0], m[1], m[2], m[3]);
                *reinterpret_cast<uint4 *>(d + 16 * q + 2 * PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
            }
        }
    }
    __device__ __forceinline__ void store(int tid, unsigned char *planes, int k_lim) const {
        if (k_lim >= 32) store_as<true>(tid, planes, k_lim);  // (workgroup-uniform)
        else store_as<false>(tid, planes, k_lim);
    }
};

// (the workgroup's tile (bx, by) and K split bz of nz come as arguments: the kernel below passes its block indices, the PAIR kernel
// further down the indices of whichever of its two products the workgroup belongs to)
template <bool A_MK, bool B_NK, int EPI, int TN>
__device__ __forceinline__ void gemm128_bf16x3_body(const GemmArgs &G, const int bx, const int by, const int bz, const int nz,
                                                    unsigned char *lds_b) {
    constexpr int TM = 128, KC = 32, T = 256, NT = TN / 64;
    constexpr int PLANE_A = TM * kBfPitch, PLANE_B = TN * kBfPitch;
    unsigned char *sA = lds_b, *sB = lds_b + 3 * PLANE_A;
    int *s_idx = reinterpret_cast<int *>(lds_b + 3 * (PLANE_A + PLANE_B));  // [kper_] row ids of the K range (B_KN with b_idx)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = bx * TM, n0 = by * TN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * (TN / 2);
    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int kper_ = ((G.K + nz - 1) / nz + KC - 1) / KC * KC;
    __shared__ int s_red[16];
    int k_cut;
    if (gemm_depth_cut(G, m0, TM, n0, s_red, k_cut)) return;  // (workgroup-uniform)
    const int k_lo = bz * kper_, k_hi = min(min(G.K, k_cut), k_lo + kper_);
    if constexpr (!B_NK) {
        if (G.b_idx) {
            // (padded to whole chunks with the last id: the loads of a partial chunk need no clamp)
            for (int k = k_lo + tid; k < k_lo + (k_hi - k_lo + KC - 1) / KC * KC; k += T) s_idx[k - k_lo] = (int)G.b_idx[min(k, k_hi - 1)];
            __syncthreads();
        }
    }
    typedef Bf16Stage<A_MK, TM> StageA;
    typedef Bf16Stage<B_NK, TN> StageB;
    StageA ra;
    StageB rb;
    // Operand addresses = descriptor base + 32-bit offsets (the launcher checks that both operands span less than 4 GB).
    // Indices are clamped into the operands (rows beyond M / N feed outputs nobody stores; k beyond the range is zeroed at the
    // LDS store); B_NK: the (gathered) rows a lane stages are the same for every K chunk: resolved once.
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.A), 0, 0xFFFFFFFFu, 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G.B), 0, 0xFFFFFFFFu, 0x00020000);
    int voff_a[StageA::NROWS], voff_b[StageB::NROWS];
#pragma unroll
    for (int e = 0; e < StageA::NROWS; ++e) {
        const int64_t m = min(m0 + StageA::row_of(tid, e), G.M - 1);
        voff_a[e] = (int)((A_MK ? m * G.lda + StageA::k_of(tid, e) : m) * 4);
    }
#pragma unroll
    for (int e = 0; e < StageB::NROWS; ++e) {
        const int64_t sel = min(n0 + StageB::row_of(tid, e), G.N - 1);
        if constexpr (B_NK) voff_b[e] = (int)(((G.b_idx ? G.b_idx[sel] : sel) * G.ldb + StageB::k_of(tid, e)) * 4);
        else voff_b[e] = (int)(sel * 4);
    }
    const int kk0_a = StageA::k_of(tid, 0), kk0_b = StageB::k_of(tid, 0);  // (row-contiguous operands: the wave's first k of a chunk)
    auto load_into = [&](StageA &ra, StageB &rb, int k0) {
        ra.load(rs_a, voff_a, tid, [&](int j) {
            if constexpr (A_MK) return (min(k0 + StageA::k_of(tid, j), k_hi - 4) - StageA::k_of(tid, j)) * 4;
            else return (int)((int64_t)min(k0 + kk0_a + j, k_hi - 1) * G.lda * 4);
        });
        // gathered rows: the wave's NEL row ids of the chunk are consecutive words of s_idx: fetched with 16-byte LDS reads up front
        // (one read + wait per load put NEL LDS round trips in front of every chunk's loads)
        int ids[StageB::NEL];
        if constexpr (!B_NK) {
            if (G.b_idx) {
                const int4 *ip = reinterpret_cast<const int4 *>(s_idx + (k0 - k_lo) + kk0_b);
#pragma unroll
                for (int q = 0; q < StageB::NEL / 4; ++q) {
                    const int4 w = ip[q];
                    ids[4 * q] = w.x; ids[4 * q + 1] = w.y; ids[4 * q + 2] = w.z; ids[4 * q + 3] = w.w;
                }
            }
        }
        rb.load(rs_b, voff_b, tid, [&](int j) {
            if constexpr (B_NK) return (min(k0 + StageB::k_of(tid, j), k_hi - 4) - StageB::k_of(tid, j)) * 4;
            else {
                const int64_t row = G.b_idx ? (int64_t)__builtin_amdgcn_readfirstlane(ids[j]) : (int64_t)min(k0 + kk0_b + j, k_hi - 1);
                return (int)(row * G.ldb * 4);
            }
        });
    };
    auto multiply = [&]() {
        const unsigned char *pa = sA + (wm + (lane & 31)) * kBfPitch + (lane >> 5) * 16;
        const unsigned char *pb = sB + (wn + (lane & 31)) * kBfPitch + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 fa[2][3], fb[NT][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    fa[a][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pa + p * PLANE_A + a * 32 * kBfPitch + ks * 32));
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    fb[b][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pb + p * PLANE_B + b * 32 * kBfPitch + ks * 32));
            }
            // smallest terms first; the (plane of a, plane of b) pairs kept: 0 = hi, 1 = mid, 2 = lo
            constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA[t]], fb[b][PB[t]], acc[a][b], 0, 0, 0);
        }
    };
    // (Round 4 also built a pipelined form -- two LDS stages, two register sets, the next chunk split under this chunk's matrix
    // instructions, one barrier per chunk -- for launches of one workgroup per CU: slower everywhere (the score product 19.7 ->
    // 28 us, dX 23.6 -> 35.8, dQ the same): two workgroups per CU filling each other's staging phases beat one that pipelines.)
    if (k_lo < k_hi) load_into(ra, rb, k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += KC) {
        ra.store(tid, sA, k_hi - k0);
        rb.store(tid, sB, k_hi - k0);
        __syncthreads();
        if (k0 + KC < k_hi) load_into(ra, rb, k0 + KC);  // the next chunk's loads fly under this chunk's matrix work
        multiply();
        __syncthreads();
    }
    float *Cz = G.C + (int64_t)bz * G.M * G.ldc;  // partial buffer of this K split (z = 0: C itself)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = n0 + wn + 32 * b + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + wm + 32 * a + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (m < G.M && n < G.N) {
                    const float v = acc[a][b][reg];
                    if constexpr (EPI == GEMM_STORE) Cz[(int64_t)m * G.ldc + n] = v;
                    else if constexpr (EPI == GEMM_STORE_AFFINE) Cz[(int64_t)m * G.ldc + n] = (nz > 1) ? v : G.c0 + G.c1 * v;
                    else if (v != 0.f) atomicAdd(G.C + G.c_idx[m] * G.ldc + n, v);
                }
            }
        }
}

template <bool A_MK, bool B_NK, int EPI, int TN>
__global__ __launch_bounds__(256) void gemm128_bf16x3_mfma_kernel(GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_gemm_bf[];
    gemm128_bf16x3_body<A_MK, B_NK, EPI, TN>(G, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.z, lds_gemm_bf);
}

// Two independent products in ONE launch (round 5): the backward of the bilinear models is dQ = G . X followed by dX = G^T . Q --
// two ~20 us launches of ~256 four-wave workgroups each, i.e. ONE workgroup per CU whose matrix pipe idles while it stages a chunk,
// and a kernel boundary + ramp between them.  As one grid with the two products' workgroups interleaved, every CU holds one of
// each from the start and they fill each other's staging phases.  grid: 1-D; P1 / P2: (tiles in m, tiles in n, K splits).
struct GemmPairGrid { int mx1, ny1, nz1, mx2, ny2, nz2; };
template <bool A1, bool B1, int E1, int TN1, bool A2, bool B2, int E2, int TN2>
__global__ __launch_bounds__(256) void gemm128_bf16x3_pair_kernel(GemmArgs G1, GemmArgs G2, GemmPairGrid P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_gemm_pair[];
    const int n1 = P.mx1 * P.ny1 * P.nz1, n2 = P.mx2 * P.ny2 * P.nz2, both = 2 * min(n1, n2);
    const int b = (int)blockIdx.x;
    int which, idx;  // (workgroup-uniform)
    if (b < both) { which = b & 1; idx = b >> 1; }
    else { which = n1 > n2 ? 0 : 1; idx = b - both + min(n1, n2); }
    if (which == 0) {
        if (idx < n1) gemm128_bf16x3_body<A1, B1, E1, TN1>(G1, idx % P.mx1, (idx / P.mx1) % P.ny1, idx / (P.mx1 * P.ny1), P.nz1, lds_gemm_pair);
    } else {
        if (idx < n2) gemm128_bf16x3_body<A2, B2, E2, TN2>(G2, idx % P.mx2, (idx / P.mx2) % P.ny2, idx / (P.mx2 * P.ny2), P.nz2, lds_gemm_pair);
    }
}

// out[i] = c0 + c1 * sum_z part[z][i]   (fixed order: deterministic)
template <int UNUSED = 0>  // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int64_t n,
                                                            int nz, float c0, float c1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z0 = 0; z0 < nz; z0 += 8) {  // eight splits' loads at a time (one by one they are a round trip each), added in split order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(int64_t)min(z0 + k, nz - 1) * n + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += z0 + k < nz ? v[k] : 0.f;
        }
        out[i] = c0 + c1 * s;
    }
}

// out[c_idx[m]][n] += sum_z part[z][m][n]: the scattered product goes through split-K partials and ONE atomic per element
// (pool ids may repeat and the positive triples' rows share gradient rows) instead of one atomic per element and K split.
__device__ __forceinline__ void splitk_scatter_block(const float *__restrict__ part, float *__restrict__ out,
                                                     const int64_t *__restrict__ c_idx, int M, int N, int64_t ldc, int nz, int m,
                                                     const int *__restrict__ depth = nullptr, int n_depth = 0, int *s_red = nullptr) {
    if (depth && m >= block_max_i32(depth, 0, n_depth, s_red)) return;  // a pool position nobody uses: its partials were skipped
    float *row = out + c_idx[m] * ldc;
    for (int n = threadIdx.x * 4; n < N; n += 1024) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z0 = 0; z0 < nz; z0 += 8) {  // eight splits' loads at a time, added in split order
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(part + ((int64_t)min(z0 + k, nz - 1) * M + m) * N + n);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (z0 + k < nz) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
        }
        if (a.x != 0.f) atomicAdd(row + n, a.x);
        if (a.y != 0.f) atomicAdd(row + n + 1, a.y);
        if (a.z != 0.f) atomicAdd(row + n + 2, a.z);
        if (a.w != 0.f) atomicAdd(row + n + 3, a.w);
    }
}

template <int UNUSED = 0>  // (as above)
__global__ __launch_bounds__(256) void splitk_scatter_kernel(const float *__restrict__ part, float *__restrict__ out,
                                                             const int64_t *__restrict__ c_idx, int M, int N, int64_t ldc, int nz,
                                                             const int *__restrict__ depth, int n_depth) {
    __shared__ int s_red[16];
    splitk_scatter_block(part, out, c_idx, M, N, ldc, nz, (int)blockIdx.x, depth, n_depth, s_red);
}

// These products are small (1-2 GFLOP) and short in one dimension: fill the chip by halving the tile height and / or
// splitting K (ksplit > 1: STORE epilogues go through `partials` [ksplit, M, ldc] and a fixed-order reduction;
// the atomic epilogue just accumulates).
// tail: non-null = the caller folds a split-K reduction / scatter into its next launch (GemmTail, common.h): it is described
// there instead of being launched (kind 0 when the product needed none).
template <bool A_MK, bool B_NK, int EPI>
// slices_cap > 0 (STORE epilogue): the caller's consumer adds up to slices_cap partial products itself -- C is [slices][M][ldc];
// a K split then writes its partials straight there (no partial buffer, no reduction launch) and *slices_used says how many.
// max_ks: the largest K split the plan may choose (1 = never split: a caller whose `partials` is only the permission to take
// the 128-row tiles and holds no room for [ks][M][ldc] partial products -- the ranking route).
static int launch_gemm(GemmArgs G, hipStream_t st, float *partials = nullptr, GemmTail *tail = nullptr, int want_wg = 0,
                       int slices_cap = 0, int *slices_used = nullptr, int max_ks = 8) {
    if (tail) tail->kind = 0;
    const bool to_slices = slices_cap > 0 && EPI == GEMM_STORE && slices_used;
    if (slices_used) *slices_used = 1;
    {   // 128 x 128 / 128 x 64 tiles (4 / 2 accumulators per wave) when the operands allow 16-byte loads
        static const bool off = getenv("MKB_GEMM_NO128") != nullptr;  // A/B switch
        const bool al = (((uintptr_t)G.A | (uintptr_t)G.B) & 15) == 0 && G.lda % 4 == 0 && G.ldb % 4 == 0 && G.M % 4 == 0 &&
                        G.N % 4 == 0 && G.K % 4 == 0;
        const int tiles128 = ((G.M + 127) / 128) * ((G.N + 127) / 128);
        if (!off && al && tiles128 >= 24 && (EPI == GEMM_ATOMIC_ROWS || partials)) {
            // fill ~256 CUs: prefer the narrower tile (no K split, no reduction launch) to splitting K
            const bool narrow = tiles128 < 200;
            const int tiles = narrow ? ((G.M + 127) / 128) * ((G.N + 63) / 64) : tiles128;
            int ks = 1;
            // workgroups wanted: ~1 per CU by default; the score product asks for 2 (want_wg = 500: a lone 4-wave workgroup leaves
            // each SIMD's matrix pipe idle during its staging; measured 37 -> 29 us, the split-K sum rides the loss rows anyway)
            static const int env_wg = getenv("MKB_GEMM_MIN_WG") ? atoi(getenv("MKB_GEMM_MIN_WG")) : 0;  // experiment knob
            const int min_wg = env_wg ? env_wg : (want_wg ? want_wg : 200);
            while (tiles * ks < min_wg && ks < max_ks && G.K / (ks * 2) >= 96 && (!to_slices || ks * 2 <= slices_cap)) ks *= 2;
            G.ksplit = ks;
            if (to_slices) *slices_used = ks;
            float *final_c = G.C;
            const int tn = narrow ? 64 : 128;
            const size_t lds = (size_t)2 * (128 + tn) * 33 * 4 + (size_t)(((G.K + ks - 1) / ks + 31) / 32 * 32) * 4;  // two stages + row ids
            dim3 grid((unsigned)((G.M + 127) / 128), (unsigned)((G.N + tn - 1) / tn), (unsigned)ks);
            // (two stages of a 128 x 128 tile pair are 66 KB: more than the 64 KB a kernel gets without asking)
            // the three-way bf16 split on the bf16 matrix pipe (gemm128_bf16x3_mfma_kernel) unless MKB_GEMM_BF16X3=0; read per call
            const char *bx = getenv("MKB_GEMM_BF16X3");
            // (its buffer loads address an operand with 32-bit byte offsets: every row it can touch must lie within 4 GB of the base)
            const int64_t a_span = (A_MK ? (int64_t)G.M * G.lda : (int64_t)G.K * G.lda) * 4;
            const int64_t b_rows = G.b_idx ? G.b_rows : (B_NK ? (int64_t)G.N : (int64_t)G.K);
            // (offsets are added as SIGNED 32-bit integers in the staging: operands of 2 GB and more take the fp32 kernel)
            const bool fits32 = a_span < ((int64_t)1 << 31) && b_rows > 0 && b_rows * G.ldb * 4 < ((int64_t)1 << 31);
            const bool bf16x3 = fits32 && (bx ? bx[0] == '1' : kGemmBf16x3Default);
            const size_t lds_bf = (size_t)3 * (128 + tn) * kBfPitch + (size_t)(((G.K + ks - 1) / ks + 31) / 32 * 32) * 4;
            auto launch128 = [&](auto epi_c, auto tn_c, const GemmArgs &GA) -> int {
                constexpr int E = decltype(epi_c)::value, TNv = decltype(tn_c)::value;
                static LdsOptIn grant[2];  // (per instantiation of this generic lambda and kernel form)
                if constexpr (E != GEMM_ATOMIC_ROWS) {  // (the scattered product goes through split-K partials: STORE; the direct
                                                        // atomic epilogue -- a caller without a partial buffer -- keeps the fp32 kernel)
                    if (bf16x3) {
                        auto *fn = &gemm128_bf16x3_mfma_kernel<A_MK, B_NK, E, TNv>;
                        if (int rc = grant[1].ensure(reinterpret_cast<const void *>(fn), lds_bf)) return rc;
                        hipLaunchKernelGGL(fn, grid, dim3(256), lds_bf, st, GA);
                        return MKB_OK;
                    }
                }
                auto *fn = &gemm128_f32_mfma_kernel<A_MK, B_NK, E, TNv>;
                if (int rc = grant[0].ensure(reinterpret_cast<const void *>(fn), lds)) return rc;
                hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, GA);
                return MKB_OK;
            };
            typedef std::integral_constant<int, 64> tn64_t;
            typedef std::integral_constant<int, 128> tn128_t;
            if (EPI == GEMM_ATOMIC_ROWS && partials) {  // partial products [ks, M, N], then one scattered add per element
                GemmArgs P2 = G;
                P2.C = partials; P2.ldc = G.N;
                typedef std::integral_constant<int, GEMM_STORE> store_t;
                if (int rc = narrow ? launch128(store_t{}, tn64_t{}, P2) : launch128(store_t{}, tn128_t{}, P2)) return rc;
                const int *dp = G.depth_mode == 3 ? G.depth : nullptr;
                if (tail) *tail = GemmTail{2, partials, final_c, G.c_idx, G.M, G.N, ks, G.ldc, 0, 0.f, 1.f, dp, G.n_depth};
                else hipLaunchKernelGGL(splitk_scatter_kernel<0>, dim3((unsigned)G.M), dim3(256), 0, st, partials, final_c, G.c_idx, G.M,
                                        G.N, G.ldc, ks, dp, G.n_depth);
                MKB_LAUNCH_CHECK();
                return MKB_OK;
            }
            if (ks > 1 && EPI != GEMM_ATOMIC_ROWS && !to_slices) G.C = partials;
            typedef std::integral_constant<int, EPI> epi_t;
            if (int rc = narrow ? launch128(epi_t{}, tn64_t{}, G) : launch128(epi_t{}, tn128_t{}, G)) return rc;
            if (ks > 1 && EPI != GEMM_ATOMIC_ROWS && !to_slices) {
                const int64_t n = (int64_t)G.M * G.ldc;
                const float c0 = EPI == GEMM_STORE_AFFINE ? G.c0 : 0.f, c1 = EPI == GEMM_STORE_AFFINE ? G.c1 : 1.f;
                if (tail) *tail = GemmTail{1, partials, final_c, nullptr, G.M, G.N, ks, G.ldc, n, c0, c1, nullptr, 0};
                else hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3(512), dim3(256), 0, st, partials, final_c, n, ks, c0, c1);
            }
            MKB_LAUNCH_CHECK();
            return MKB_OK;
        }
    }
    const int tiles64 = ((G.M + 63) / 64) * ((G.N + 63) / 64);
    const bool half = tiles64 < 256;
    const int tiles = half ? ((G.M + 31) / 32) * ((G.N + 63) / 64) : tiles64;
    int ks = 1;
    if (EPI == GEMM_ATOMIC_ROWS || partials || to_slices) {
        while (tiles * ks < 768 && ks < max_ks && G.K / (ks * 2) >= 128 && (!to_slices || ks * 2 <= slices_cap)) ks *= 2;
    }
    G.ksplit = ks;
    if (to_slices) *slices_used = ks;
    float *final_c = G.C;
    if (ks > 1 && EPI != GEMM_ATOMIC_ROWS && !to_slices) G.C = partials;
    dim3 grid((unsigned)((G.M + (half ? 31 : 63)) / (half ? 32 : 64)), (unsigned)((G.N + 63) / 64), (unsigned)ks);
    if (half) hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_MK, B_NK, EPI, 32>), grid, dim3(128), 0, st, G);
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_MK, B_NK, EPI, 64>), grid, dim3(256), 0, st, G);
    if (ks > 1 && EPI != GEMM_ATOMIC_ROWS && !to_slices) {
        const int64_t n = (int64_t)G.M * G.ldc;
        const float c0 = EPI == GEMM_STORE_AFFINE ? G.c0 : 0.f, c1 = EPI == GEMM_STORE_AFFINE ? G.c1 : 1.f;
        hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3(512), dim3(256), 0, st, partials, final_c, n, ks, c0, c1);
    }
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

// The backward pair of the bilinear models as ONE launch (gemm128_bf16x3_pair_kernel): dQ = G . X (A k-contiguous, B gathered rows,
// STORE into up to `slices_cap` dQ slices) and the scattered dX = G^T . Q (through split-K partials, its tail -- one atomic per
// element -- described in *x_tail or launched).  Plans each product exactly as launch_gemm would (tile width, K split); *done =
// false when either does not qualify for the 128-row bf16 tiles (the caller then launches them one by one).
static int launch_gemm_bwd_pair(GemmArgs GQ, int slices_cap, int *slices_used, GemmArgs GX, float *partials, GemmTail *x_tail,
                                hipStream_t st, bool *done) {
    *done = false;
    const bool off = getenv("MKB_GEMM_NO128") != nullptr || getenv("MKB_GEMM_NO_PAIR") != nullptr;  // A/B switches (read per call: the tests flip them within one process)
    const char *bx = getenv("MKB_GEMM_BF16X3");
    if (off || !(bx ? bx[0] == '1' : kGemmBf16x3Default) || !partials || !slices_used || slices_cap < 1) return MKB_OK;
    struct Plan { bool ok, narrow; int ks, tn, mx, ny; size_t lds; };
    auto plan = [&](const GemmArgs &G, bool a_mk, bool b_nk, int cap) {
        Plan P{false, false, 1, 128, 0, 0, 0};
        const bool al = (((uintptr_t)G.A | (uintptr_t)G.B) & 15) == 0 && G.lda % 4 == 0 && G.ldb % 4 == 0 && G.M % 4 == 0 && G.N % 4 == 0 &&
                        G.K % 4 == 0;
        const int tiles128 = ((G.M + 127) / 128) * ((G.N + 127) / 128);
        const int64_t a_span = (a_mk ? (int64_t)G.M * G.lda : (int64_t)G.K * G.lda) * 4;
        const int64_t b_rows = G.b_idx ? G.b_rows : (b_nk ? (int64_t)G.N : (int64_t)G.K);
        const bool fits32 = a_span < ((int64_t)1 << 31) && b_rows > 0 && b_rows * G.ldb * 4 < ((int64_t)1 << 31);
        if (!al || tiles128 < 24 || !fits32) return P;
        P.narrow = tiles128 < 200;
        P.tn = P.narrow ? 64 : 128;
        const int tiles = P.narrow ? ((G.M + 127) / 128) * ((G.N + 63) / 64) : tiles128;
        static const int env_wg = getenv("MKB_GEMM_MIN_WG") ? atoi(getenv("MKB_GEMM_MIN_WG")) : 0;
        // (short rows -- DistMult's 1000 floats -- leave few tiles: split K further there: 26.5 -> 23.4 us for its pair; the
        // 2000-float rows of ComplEx lose with more splits, 42 -> 49 us: their partials are what the extra workgroups move)
        const int min_wg = env_wg ? env_wg : (G.N <= 1024 ? 400 : 200);
        while (tiles * P.ks < min_wg && P.ks < 8 && G.K / (P.ks * 2) >= 96 && (cap <= 0 || P.ks * 2 <= cap)) P.ks *= 2;
        P.mx = (G.M + 127) / 128; P.ny = (G.N + P.tn - 1) / P.tn;
        P.lds = (size_t)3 * (128 + P.tn) * kBfPitch + (size_t)(((G.K + P.ks - 1) / P.ks + 31) / 32 * 32) * 4;
        P.ok = true;
        return P;
    };
    const Plan pq = plan(GQ, true, false, slices_cap), px = plan(GX, false, false, 0);
    if (!pq.ok || !px.ok) return MKB_OK;
    GQ.ksplit = pq.ks;
    *slices_used = pq.ks;
    GemmArgs X2 = GX;  // partial products [ks, M, N], then one scattered add per element (the tail)
    X2.C = partials; X2.ldc = GX.N; X2.ksplit = px.ks;
    const GemmPairGrid grid{pq.mx, pq.ny, pq.ks, px.mx, px.ny, px.ks};
    const size_t lds = pq.lds > px.lds ? pq.lds : px.lds;
    const unsigned blocks = (unsigned)(pq.mx * pq.ny * pq.ks + px.mx * px.ny * px.ks);
    auto go = [&](auto tq, auto tx) -> int {
        constexpr int TQ = decltype(tq)::value, TX = decltype(tx)::value;
        auto *fn = &gemm128_bf16x3_pair_kernel<true, false, GEMM_STORE, TQ, false, false, GEMM_STORE, TX>;
        static LdsOptIn grant;  // (per instantiation of this generic lambda)
        if (int rc = grant.ensure(reinterpret_cast<const void *>(fn), lds)) return rc;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, st, GQ, X2, grid);
        return MKB_OK;
    };
    typedef std::integral_constant<int, 64> t64;
    typedef std::integral_constant<int, 128> t128;
    int rc;
    if (pq.narrow) rc = px.narrow ? go(t64{}, t64{}) : go(t64{}, t128{});
    else rc = px.narrow ? go(t128{}, t64{}) : go(t128{}, t128{});
    if (rc) return rc;
    const int *dp = GX.depth_mode == 3 ? GX.depth : nullptr;
    if (x_tail) *x_tail = GemmTail{2, partials, GX.C, GX.c_idx, GX.M, GX.N, px.ks, GX.ldc, 0, 0.f, 1.f, dp, GX.n_depth};
    else hipLaunchKernelGGL(splitk_scatter_kernel<0>, dim3((unsigned)GX.M), dim3(256), 0, st, partials, GX.C, GX.c_idx, GX.M, GX.N, GX.ldc,
                            px.ks, dp, GX.n_depth);
    MKB_LAUNCH_CHECK();
    *done = true;
    return MKB_OK;
}

}  // namespace mkb

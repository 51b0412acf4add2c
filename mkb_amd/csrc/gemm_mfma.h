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

struct GemmArgs {
    const float *A, *B;
    float *C;
    const int64_t *b_idx;   // optional row indirection of B (B_NK: indexed by n; B_KN: indexed by k)
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
            const int m = min(m0 + r, G.M - 4), k = min(k0 + kk, A_MK ? k_hi - 4 : k_hi - 1);
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

// out[i] = c0 + c1 * sum_z part[z][i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int64_t n,
                                                            int nz, float c0, float c1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += part[(int64_t)z * n + i];
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
        for (int z = 0; z < nz; ++z) {
            const float4 v = *reinterpret_cast<const float4 *>(part + ((int64_t)z * M + m) * N + n);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (a.x != 0.f) atomicAdd(row + n, a.x);
        if (a.y != 0.f) atomicAdd(row + n + 1, a.y);
        if (a.z != 0.f) atomicAdd(row + n + 2, a.z);
        if (a.w != 0.f) atomicAdd(row + n + 3, a.w);
    }
}

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
static int launch_gemm(GemmArgs G, hipStream_t st, float *partials = nullptr, GemmTail *tail = nullptr, int want_wg = 0) {
    if (tail) tail->kind = 0;
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
            while (tiles * ks < min_wg && ks < 8 && G.K / (ks * 2) >= 96) ks *= 2;
            G.ksplit = ks;
            float *final_c = G.C;
            const int tn = narrow ? 64 : 128;
            const size_t lds = (size_t)2 * (128 + tn) * 33 * 4 + (size_t)(((G.K + ks - 1) / ks + 31) / 32 * 32) * 4;  // two stages + row ids
            dim3 grid((unsigned)((G.M + 127) / 128), (unsigned)((G.N + tn - 1) / tn), (unsigned)ks);
            // (two stages of a 128 x 128 tile pair are 66 KB: more than the 64 KB a kernel gets without asking)
            auto launch128 = [&](auto epi_c, auto tn_c, const GemmArgs &GA) -> int {
                constexpr int E = decltype(epi_c)::value, TNv = decltype(tn_c)::value;
                static LdsOptIn grant;  // (one per instantiation of this generic lambda)
                auto *fn = &gemm128_f32_mfma_kernel<A_MK, B_NK, E, TNv>;
                if (int rc = grant.ensure(reinterpret_cast<const void *>(fn), lds)) return rc;
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
                else hipLaunchKernelGGL(splitk_scatter_kernel, dim3((unsigned)G.M), dim3(256), 0, st, partials, final_c, G.c_idx, G.M,
                                        G.N, G.ldc, ks, dp, G.n_depth);
                MKB_LAUNCH_CHECK();
                return MKB_OK;
            }
            if (ks > 1 && EPI != GEMM_ATOMIC_ROWS) G.C = partials;
            typedef std::integral_constant<int, EPI> epi_t;
            if (int rc = narrow ? launch128(epi_t{}, tn64_t{}, G) : launch128(epi_t{}, tn128_t{}, G)) return rc;
            if (ks > 1 && EPI != GEMM_ATOMIC_ROWS) {
                const int64_t n = (int64_t)G.M * G.ldc;
                const float c0 = EPI == GEMM_STORE_AFFINE ? G.c0 : 0.f, c1 = EPI == GEMM_STORE_AFFINE ? G.c1 : 1.f;
                if (tail) *tail = GemmTail{1, partials, final_c, nullptr, G.M, G.N, ks, G.ldc, n, c0, c1, nullptr, 0};
                else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(512), dim3(256), 0, st, partials, final_c, n, ks, c0, c1);
            }
            MKB_LAUNCH_CHECK();
            return MKB_OK;
        }
    }
    const int tiles64 = ((G.M + 63) / 64) * ((G.N + 63) / 64);
    const bool half = tiles64 < 256;
    const int tiles = half ? ((G.M + 31) / 32) * ((G.N + 63) / 64) : tiles64;
    int ks = 1;
    if (EPI == GEMM_ATOMIC_ROWS || partials) {
        while (tiles * ks < 768 && ks < 8 && G.K / (ks * 2) >= 128) ks *= 2;
    }
    G.ksplit = ks;
    float *final_c = G.C;
    if (ks > 1 && EPI != GEMM_ATOMIC_ROWS) G.C = partials;
    dim3 grid((unsigned)((G.M + (half ? 31 : 63)) / (half ? 32 : 64)), (unsigned)((G.N + 63) / 64), (unsigned)ks);
    if (half) hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_MK, B_NK, EPI, 32>), grid, dim3(128), 0, st, G);
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_MK, B_NK, EPI, 64>), grid, dim3(256), 0, st, G);
    if (ks > 1 && EPI != GEMM_ATOMIC_ROWS) {
        const int64_t n = (int64_t)G.M * G.ldc;
        const float c0 = EPI == GEMM_STORE_AFFINE ? G.c0 : 0.f, c1 = EPI == GEMM_STORE_AFFINE ? G.c1 : 1.f;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(512), dim3(256), 0, st, partials, final_c, n, ks, c0, c1);
    }
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

}  // namespace mkb

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
};

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
    const int k_lo = blockIdx.z * kper_, k_hi = min(G.K, k_lo + kper_);

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

// out[i] = c0 + c1 * sum_z part[z][i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int64_t n,
                                                            int nz, float c0, float c1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += part[(int64_t)z * n + i];
        out[i] = c0 + c1 * s;
    }
}

// These products are small (1-2 GFLOP) and short in one dimension: fill the chip by halving the tile height and / or
// splitting K (ksplit > 1: STORE epilogues go through `partials` [ksplit, M, ldc] and a fixed-order reduction;
// the atomic epilogue just accumulates).
template <bool A_MK, bool B_NK, int EPI>
static int launch_gemm(GemmArgs G, hipStream_t st, float *partials = nullptr) {
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

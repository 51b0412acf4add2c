// pool_fwd2: forward of the pooled block for long rows (256 <= units <= 1024, units % 4 == 0): S[i][p] = score of batch row i
// against pool position p, for the (i, p) pairs the batch uses (reference: the negative forward of compose/pipeline.py:230-232,
// i.e. models/rotate.py:83-97, transe.py:65-76, protate.py:74-93 over the shared pool of sampling/negative_sampling.py:166).
//
// Round-2's forward (pool_fwd_kernel) gave each lane 4 units of 8 rows, so a row was spread over 4 waves and every
// position paid a transposed 8-value wave reduction PLUS a cross-wave LDS combine: 31 of its 135 instructions per
// position and tile, a barrier per 16 positions, 0.15 of the fp32 peak.  Here
//   * a WAVE holds R whole rows: lane l owns units [256 c + 4 l, +4) for c < VPL of each (RotatE hidden 1000: 16 complex
//     dims = 32 floats per row and lane), so a (row, position) score is complete after ONE wave reduction of R values
//     (10 instructions for R = 4, amortised over R x 4 VPL pair terms per lane) and is stored directly: no cross-wave
//     combine, no partial buffers, no split over dims;
//   * the candidate rows are staged ONCE per workgroup in LDS by the DMA path (global_load_lds_dwordx4: 1 KB per wave
//     instruction, no staging registers, no ds_write) in a two-stage ring of G positions, one barrier per stage, and each
//     wave reads its lanes' 16-byte pieces with conflict-free ds_read_b128 -- 4 waves x R rows share every staged row;
//   * sparsity as before: the positions a workgroup's rows use (within its interleaved position slice) are compacted
//     into an LDS list with one bit per row; a wave skips positions none of its rows use, and rows inside a wave that do
//     not use a position are skipped with scalar branches (or computed and not stored when most do).
// VALU-bound: RotatE 5 packed ops + 2 v_sqrt per two complex dims and pair.
#pragma once
#include "score_pool_kernels.h"

namespace mkb {

constexpr int kF2Waves = 8;     // waves per workgroup
constexpr int kF2Group = 2;     // positions per LDS stage (one barrier per stage)
constexpr int kF2Stages = 4;    // ring depth: a stage is requested kF2Stages - 1 stages (2-3 us of pair math) before it is read

typedef __attribute__((address_space(3))) void f2_lptr_t;

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + 16 * lane (wave-uniform base in
// M0).  Inline assembly, not __builtin_amdgcn_global_load_lds: the compiler tracks the builtin as an LDS write and puts an
// s_waitcnt vmcnt(0) in front of EVERY later ds_read (seen in the ISA of the first version of this kernel: the whole request
// ring drained at each position, 78 us for the launch); with the asm form the only waits are the counted ones below.
__device__ __forceinline__ void lds_dma_16(const void *gsrc, void *lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(f2_lptr_t *)lds_dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// s_waitcnt vmcnt(N) lgkmcnt(0) for a run-time N <= 8 (the instruction takes an immediate).  vmcnt also counts this wave's
// score stores; loads complete in order, so "at most N outstanding" still implies that all but the newest N REQUESTS landed.
__device__ __forceinline__ void wait_vm_at_most(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
}


// totals of R per-lane values: on return lane (16 * t) .. (16 * t + 15) hold the wave total of value t (R = 4), or the
// pair (t0, t1) of reduce8_wave (R = 8)
__device__ __forceinline__ float reduce4_wave(const float (&v)[4]) {
    float w[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // lanes 0-31 keep values 0-1, lanes 32-63 keep values 2-3
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 2]), false, false);
        w[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[0]), __float_as_uint(w[1]), false, false);
    float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // even 16-lane rows: value 0 / 2, odd rows: value 1 / 3
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x128, 0xf, 0xf, false));
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x141, 0xf, 0xf, false));
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0xB1, 0xf, 0xf, false));
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x4E, 0xf, 0xf, false));
    return t;
}

// R = 2: lanes 0-31 hold the total of value 0, lanes 32-63 of value 1
__device__ __forceinline__ float reduce2_wave(const float (&v)[2]) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
    float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x128, 0xf, 0xf, false));
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x141, 0xf, 0xf, false));
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0xB1, 0xf, 0xf, false));
    t += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(t), 0x4E, 0xf, 0xf, false));
    return t;
}

template <int MODEL, bool HEAD, int VPL, int R>
__global__ __launch_bounds__(kF2Waves * 64) void pool_fwd2_kernel(PoolArgs A) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int NW = kF2Waves, WG = NW * 64, ROWS = NW * R, G = kF2Group;
    static_assert(R == 2 || R == 4, "rows per wave");
    // interleaved position slice of this workgroup, as in pool_fwd_kernel: p = slice + nslices * i
    const int nsl = gridDim.y, sl = blockIdx.y;
    const int Pn = (A.P - sl + nsl - 1) / nsl;
    const int Pcap = (A.P + nsl - 1) / nsl;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_f2[];
    const int pieces = (int)((A.De * 4 + 1023) >> 10);  // 1 KB DMA pieces per candidate row
    const int pitch = pieces << 10;
    unsigned char *ring = lds_f2;                                            // [stages][G][pitch]
    int *s_row = reinterpret_cast<int *>(lds_f2 + (size_t)kF2Stages * G * pitch);  // entity id per active position
    int *s_pos = s_row + Pcap;                                               // pool position
    unsigned *s_mask = reinterpret_cast<unsigned *>(s_pos + Pcap);           // bit r: row r of the workgroup uses it
    __shared__ int s_wave_cnt[NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * ROWS, iw = i0 + wave * R;
    const int NU = CP ? A.d : (int)A.De;
    const int64_t im_off = CP ? (int64_t)A.d : 0;  // floats between the two halves of a complex row

    // this wave's R query rows: whole rows, 4 VPL units per lane (out-of-range units hold 0)
    float q0[R][VPL][4], q1[R][VPL][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float *qrow = A.Q + (int64_t)min(iw + r, A.B - 1) * A.De;
#pragma unroll
        for (int c = 0; c < VPL; ++c) {
            const int u = c * 256 + lane * 4;
            const bool ok = u < NU && iw + r < A.B;
            const float4 a = *reinterpret_cast<const float4 *>(qrow + (ok ? u : 0));
            const float4 b = CP ? *reinterpret_cast<const float4 *>(qrow + im_off + (ok ? u : 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
            q0[r][c][0] = ok ? a.x : 0.f; q0[r][c][1] = ok ? a.y : 0.f; q0[r][c][2] = ok ? a.z : 0.f; q0[r][c][3] = ok ? a.w : 0.f;
            q1[r][c][0] = ok ? b.x : 0.f; q1[r][c][1] = ok ? b.y : 0.f; q1[r][c][2] = ok ? b.z : 0.f; q1[r][c][3] = ok ? b.w : 0.f;
        }
    }
    const bool tail_ok = (VPL - 1) * 256 + lane * 4 < NU;  // this lane's units of the last chunk exist

    // occurrence counts of the batch's entities for the row backward (fire-and-forget atomics of the first position slice)
    if (A.occ && sl == 0) {
        if (tid < ROWS && i0 + tid < A.B) {
            atomicAdd(A.occ + A.occ_sample[3 * (int64_t)(i0 + tid)], 1);
            atomicAdd(A.occ + A.occ_sample[3 * (int64_t)(i0 + tid) + 2], 1);
        }
        for (int p = blockIdx.x * WG + tid; p < A.P; p += gridDim.x * WG) atomicAdd(A.occ + A.pool[p], 1);
    }

    // positions used by at least one row of the workgroup, compacted into LDS
    MKB_TRACE_T(tr_t0);
    MKB_TRACE_ONLY(unsigned long long tr_wait = 0;)
    int n_act = 0;
    for (int base = 0; base < Pn; base += WG) {
        const int p = sl + (base + tid) * nsl;
        unsigned m_own = 0;
        if (base + tid < Pn) {
#pragma unroll 4
            for (int r = 0; r < ROWS; ++r) {
                if (i0 + r < A.B) {
                    const unsigned c = A.cnt[(int64_t)(i0 + r) * A.P + p];
                    m_own |= c ? (1u << r) : 0u;
                    // entries no row uses are defined as 0 (the pair loop below never visits them): no memset launch
                    if (!c) A.S[(int64_t)(i0 + r) * A.P + p] = 0.f;
                }
            }
        }
        int tot;
        const int slot = n_act + wg_compact_slot<NW>(m_own != 0, s_wave_cnt, &tot);
        if (m_own != 0) {
            s_pos[slot] = p;
            s_mask[slot] = m_own;
            s_row[slot] = (int)A.pool[p];
        }
        n_act += tot;
        __syncthreads();
    }
    if (n_act == 0) return;  // (workgroup-uniform)
    MKB_TRACE_T(tr_t1);

    // stage st = positions [st * G, st * G + G) of the list; its G * pieces DMA pieces are dealt to the waves, the SAME
    // number (npw) to each -- a wave with fewer real pieces re-requests its last one -- so that "all but the newest
    // (kF2Stages - 2) stages of my requests have landed" is one counted s_waitcnt
    const int n_stages = (n_act + G - 1) / G;
    const int npw = (G * pieces + NW - 1) / NW;
    const int row_bytes = (int)A.De * 4;
    auto issue_stage = [&](int st) {
        unsigned char *buf = ring + (size_t)(st % kF2Stages) * G * pitch;
        for (int k = 0; k < npw; ++k) {
            const int e = min(wave * npw + k, G * pieces - 1);
            const int g = e / pieces, piece = e - g * pieces;
            const int j = min(st * G + g, n_act - 1);  // (a short last stage re-stages the last position: never read)
            const float *row = A.ent + (int64_t)__builtin_amdgcn_readfirstlane(s_row[j]) * A.De;
            const int byte = min(piece * 1024 + lane * 16, row_bytes - 16);  // lanes past the row end re-read its last 16 bytes
            lds_dma_16(reinterpret_cast<const unsigned char *>(row) + byte, buf + (size_t)g * pitch + (size_t)piece * 1024);
        }
    };
    // requests run kF2Stages - 1 stages ahead of the reads (stages past the end are requested too -- clamped, never read -- so
    // that the count of outstanding requests per wave is the same in every iteration)
#pragma unroll
    for (int a = 0; a < kF2Stages - 1; ++a) issue_stage(a);
    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;
    const unsigned wmask = ((1u << R) - 1u) << (wave * R);
    for (int st = 0; st < n_stages; ++st) {
        // my pieces of stage st have landed (all but the newest kF2Stages - 2 stages of my requests); the barrier publishes
        // every wave's pieces and tells that everybody is done reading stage st - 1, whose buffer stage st + kF2Stages - 1
        // goes into
        MKB_TRACE_ONLY(const unsigned long long tw0 = __builtin_readcyclecounter();)
        wait_vm_at_most((kF2Stages - 2) * npw);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MKB_TRACE_ONLY(tr_wait += __builtin_readcyclecounter() - tw0;)
#ifndef MKB_F2_NO_DMA
        issue_stage(st + kF2Stages - 1);
#endif
        const unsigned char *buf = ring + (size_t)(st % kF2Stages) * G * pitch;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int j = st * G + g;
            if (j >= n_act) break;
            const unsigned m = (__builtin_amdgcn_readfirstlane(s_mask[j]) & wmask) >> (wave * R);
            if (m == 0u) continue;  // none of this wave's rows uses the position
            const float *xs = reinterpret_cast<const float *>(buf + (size_t)g * pitch);
            float x0[VPL][4], x1[VPL][4];
#pragma unroll
            for (int c = 0; c < VPL; ++c) {
                const float4 a = *reinterpret_cast<const float4 *>(xs + c * 256 + lane * 4);
                const float4 b = CP ? *reinterpret_cast<const float4 *>(xs + im_off + c * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const bool ok = c < VPL - 1 || tail_ok;  // lanes past the row end of the last chunk read the other half: 0
                x0[c][0] = ok ? a.x : 0.f; x0[c][1] = ok ? a.y : 0.f; x0[c][2] = ok ? a.z : 0.f; x0[c][3] = ok ? a.w : 0.f;
                x1[c][0] = ok ? b.x : 0.f; x1[c][1] = ok ? b.y : 0.f; x1[c][2] = ok ? b.z : 0.f; x1[c][3] = ok ? b.w : 0.f;
            }
            float part[R];
            auto rows = [&](auto dense_c) {
                constexpr bool DENSE = decltype(dense_c)::value;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    part[r] = 0.f;
                    if (DENSE || (m & (1u << r))) {  // out-of-range units hold q = x = 0 and contribute exactly 0
                        if constexpr (CP) {
                            f2 acc = pair_term_cmod2(f2{q0[r][0][0], q0[r][0][1]}, f2{q1[r][0][0], q1[r][0][1]}, f2{x0[0][0], x0[0][1]},
                                                     f2{x1[0][0], x1[0][1]});
                            acc += pair_term_cmod2(f2{q0[r][0][2], q0[r][0][3]}, f2{q1[r][0][2], q1[r][0][3]}, f2{x0[0][2], x0[0][3]},
                                                   f2{x1[0][2], x1[0][3]});
#pragma unroll
                            for (int c = 1; c < VPL; ++c) {
                                acc += pair_term_cmod2(f2{q0[r][c][0], q0[r][c][1]}, f2{q1[r][c][0], q1[r][c][1]},
                                                       f2{x0[c][0], x0[c][1]}, f2{x1[c][0], x1[c][1]});
                                acc += pair_term_cmod2(f2{q0[r][c][2], q0[r][c][3]}, f2{q1[r][c][2], q1[r][c][3]},
                                                       f2{x0[c][2], x0[c][3]}, f2{x1[c][2], x1[c][3]});
                            }
                            part[r] = acc.x + acc.y;
                        } else {
#pragma unroll
                            for (int c = 0; c < VPL; ++c)
#pragma unroll
                                for (int v = 0; v < 4; ++v) part[r] += pair_term_real<MODEL, HEAD>(q0[r][c][v], x0[c][v], A.kd);
                        }
                    }
                }
            };
#ifdef MKB_F2_NO_MATH  // (A/B builds of tools/kbench.py: what the launch costs without the pair math)
#pragma unroll
            for (int r = 0; r < R; ++r) part[r] = x0[0][r & 3] + x1[VPL - 1][r & 3] + q0[r][0][0];
#else
            if (__builtin_popcount(m) >= (R > 2 ? R - 1 : R)) rows(std::true_type{});
            else rows(std::false_type{});
#endif
            const int p = __builtin_amdgcn_readfirstlane(s_pos[j]);
            if constexpr (R == 4) {
                float t = reduce4_wave(part);
                const int r = lane >> 4;
                if ((lane & 15) == 0 && (m & (1u << r))) {
                    if constexpr (MODEL == MKB_PROTATE) t *= modulus;  // gamma - modulus * sum (protate.py:91)
                    A.S[(int64_t)(iw + r) * A.P + p] = A.c0 + A.c1 * t;
                }
            } else {
                float t = reduce2_wave(part);
                const int r = lane >> 5;
                if ((lane & 31) == 0 && (m & (1u << r))) {
                    if constexpr (MODEL == MKB_PROTATE) t *= modulus;
                    A.S[(int64_t)(iw + r) * A.P + p] = A.c0 + A.c1 * t;
                }
            }
        }
    }
    // requests past the last stage may still be in flight: they must land before the workgroup gives its LDS back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MKB_TRACE_OUT(A, 0, tr_t0, tr_t1, wall_clock64(), (unsigned long long)n_act | ((tr_wait >> 6) << 16));
}

template <int MODEL, bool HEAD, int VPL, int R>
static int launch_fwd2_cfg(const PoolLaunch &L, const PoolArgs &A, hipStream_t st) {
    constexpr int ROWS = kF2Waves * R;
    const int pieces = (int)((A.De * 4 + 1023) >> 10);
    const size_t lds = (size_t)kF2Stages * kF2Group * pieces * 1024 + (size_t)3 * ((A.P + L.fwd2_slices - 1) / L.fwd2_slices) * 4;
    static size_t lds_ok = 0;  // per instantiation: opt in to more than 64 KB of dynamic LDS (static LDS counts against 160 KB)
    if (lds > 64 * 1024 && lds > lds_ok) {
        MKB_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pool_fwd2_kernel<MODEL, HEAD, VPL, R>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_ok = lds;
    }
    dim3 grid((unsigned)((A.B + ROWS - 1) / ROWS), (unsigned)L.fwd2_slices);
    hipLaunchKernelGGL((pool_fwd2_kernel<MODEL, HEAD, VPL, R>), grid, dim3(kF2Waves * 64), lds, st, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int launch_fwd2(const PoolLaunch &L, const PoolArgs &A, hipStream_t st) {
    constexpr bool CP = ModelTraits<MODEL>::cplx_pair;
    constexpr int R = CP ? 2 : 4;  // a complex unit is two floats: 2 rows x 32 floats or 4 rows x 16 floats per lane (4 waves / SIMD)
    const int NU = CP ? A.d : (int)A.De;
    const int vpl = (NU + 255) / 256;
    switch (vpl) {
        case 1: return launch_fwd2_cfg<MODEL, HEAD, 1, R>(L, A, st);
        case 2: return launch_fwd2_cfg<MODEL, HEAD, 2, R>(L, A, st);
        case 3: return launch_fwd2_cfg<MODEL, HEAD, 3, R>(L, A, st);
        case 4: return launch_fwd2_cfg<MODEL, HEAD, 4, R>(L, A, st);
    }
    return set_error(MKB_ERR_UNSUPPORTED, "pool_fwd2: %d units per row", NU);
}

}  // namespace mkb

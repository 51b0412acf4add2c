// The pool draw of the negative sampler (numpy legacy MT19937 + masked rejection + the per-batch helper tables) as a
// device function of ONE workgroup of NT lanes, so that it can run as its own kernel (mkb_sampler_generate) or ride as
// block 0 of another launch (mkb_adam_rows_step's draw_ahead: the next step's pool is drawn in the shadow of the
// optimizer kernel, one launch and ~13 us of serial latency fewer per training step).
#pragma once
#include "common.h"

namespace mkb {

constexpr int MT_N = 624, MT_M = 397;

struct DrawArgs {
    uint32_t *mt;          // [624] generator state (in / out)
    int32_t *mtpos;        // [1]
    uint32_t *mt_prev;     // null, or [625]: the state before this draw is saved here (words 0..623, position in 624)
    uint32_t rng;          // n_entity - 1
    int P, P2;             // 2K and the next power of two
    int64_t *pool;         // [P] sampler-owned copy
    int64_t *pool2;        // null, or the caller's [P] output
    uint8_t *lastflag;     // [P]
    int32_t *sorted_val, *sorted_pos;  // [P2]
};
inline size_t draw_lds_bytes(int P, int P2) { return (size_t)P2 * 8 + (size_t)P * 4; }

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// exclusive prefix sum of a 0/1 flag over an NT-lane block; returns rank, *total = block total
template <int NT>
__device__ __forceinline__ int block_scan_flag(bool flag, int *wave_tot, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wave_tot[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const int c = wave_tot[w];
        if (w < wave) base += c;
        tot += c;
    }
    *total = tot;
    return base + in_wave;
}

// skey: dynamic LDS of draw_lds_bytes(P, P2) bytes, 8-byte aligned.  Every lane of the NT-lane workgroup must call it.
template <int NT>
__device__ __forceinline__ void pool_draw_body(const DrawArgs &D, unsigned long long *skey) {
    static_assert(NT >= 256 && NT % 64 == 0, "the twist phases need >= 227 lanes");
    const int P = D.P, P2 = D.P2;
    const uint32_t rng = D.rng;
    int64_t *__restrict__ pool = D.pool;
    int64_t *__restrict__ pool2 = D.pool2;
    uint32_t *pool_out_l = reinterpret_cast<uint32_t *>(skey + P2);  // [P] drawn values
    __shared__ uint32_t mt[MT_N];
    __shared__ int wave_tot[16];
    __shared__ int s_newpos;
    const int tid = threadIdx.x;
    for (int i = tid; i < MT_N; i += NT) mt[i] = D.mt[i];
    int pos = D.mtpos[0];
    if (D.mt_prev) {  // state before this draw (mkb_sampler_get_state while a pool is drawn ahead)
        for (int i = tid; i < MT_N; i += NT) D.mt_prev[i] = D.mt[i];
        if (tid == 0) D.mt_prev[MT_N] = (uint32_t)pos;
    }
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    __syncthreads();
    int have = 0;
    if (rng == 0) {  // randint(1): no stream consumption
        for (int p = tid; p < P; p += NT) { pool[p] = 0; if (pool2) pool2[p] = 0; pool_out_l[p] = 0; }
        have = P;
    }
    while (have < P) {
        if (pos == MT_N) {  // regenerate the block: three parallel phases + the last word
            uint32_t nv = 0;
            if (tid < 227) nv = mt_twist(mt[tid], mt[tid + 1], mt[tid + MT_M]);
            __syncthreads();
            if (tid < 227) mt[tid] = nv;
            __syncthreads();
            if (tid < 227) nv = mt_twist(mt[227 + tid], mt[228 + tid], mt[tid]);
            __syncthreads();
            if (tid < 227) mt[227 + tid] = nv;
            __syncthreads();
            if (tid < 169) nv = mt_twist(mt[454 + tid], mt[455 + tid], mt[227 + tid]);
            __syncthreads();
            if (tid < 169) mt[454 + tid] = nv;
            __syncthreads();
            if (tid == 0) mt[623] = mt_twist(mt[623], mt[0], mt[396]);
            __syncthreads();
            pos = 0;
        }
        const int avail = min(NT, MT_N - pos);  // raw outputs examined this round
        uint32_t v = 0;
        bool acc = false;
        if (tid < avail) {
            v = mt_temper(mt[pos + tid]) & mask;
            acc = v <= rng;
        }
        int total;
        const int rank = block_scan_flag<NT>(acc, wave_tot, &total);
        const int need = P - have;
        if (acc && rank < need) {
            pool[have + rank] = (int64_t)v;
            if (pool2) pool2[have + rank] = (int64_t)v;
            pool_out_l[have + rank] = v;
        }
        if (tid == 0) s_newpos = pos + avail;
        __syncthreads();
        if (acc && rank == need - 1) s_newpos = pos + tid + 1;  // word that produced the last needed draw
        __syncthreads();
        pos = s_newpos;
        have += (total < need) ? total : need;
        __syncthreads();
    }
    for (int i = tid; i < MT_N; i += NT) D.mt[i] = mt[i];
    if (tid == 0) D.mtpos[0] = pos;
    __syncthreads();
    // ---- per-batch helpers for the row filter, all in LDS ------------------------------------------------
    // keys[e] = entity << 13 | position, sorted ascending (bitonic): equal entities are adjacent, positions ascending
    for (int e = tid; e < P2; e += NT)
        skey[e] = e < P ? (((unsigned long long)(pool_out_l[e])) << 13) | (unsigned)e : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int e = tid; e < P2; e += NT) {
                const int partner = e ^ j;
                if (partner > e) {
                    const unsigned long long a = skey[e], b = skey[partner];
                    const bool up = (e & k) == 0;
                    if ((a > b) == up) { skey[e] = b; skey[partner] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < P2; e += NT) {
        const unsigned long long kk = skey[e];
        const bool real = kk != ~0ull;
        D.sorted_val[e] = real ? (int32_t)(kk >> 13) : INT32_MAX;
        D.sorted_pos[e] = real ? (int32_t)(kk & 8191u) : -1;
        // lastflag[p]: no later position holds the same entity == next sorted key has a different entity
        if (real) {
            const unsigned long long nx = (e + 1 < P2) ? skey[e + 1] : ~0ull;
            D.lastflag[kk & 8191u] = (nx == ~0ull || (nx >> 13) != (kk >> 13)) ? 1 : 0;
        }
    }
}

}  // namespace mkb

// sampler.hip: hand the next pool draw of `s` to another launch (fills *D, returns the dynamic LDS it needs); the next
// mkb_sampler_generate then skips its own draw kernel.  false: nothing to piggyback (a drawn pool is already waiting).
struct mkb_sampler;
namespace mkb {
bool sampler_draw_ahead(mkb_sampler *s, DrawArgs *D, size_t *lds_bytes);
}

// The pool draw of the negative sampler (numpy legacy MT19937 + masked rejection + the per-batch helper tables) as a
// device function of ONE workgroup of NT lanes, so that it can run as its own kernel (mkb_sampler_generate) or ride as
// block 0 of another launch (mkb_adam_rows_step's draw_ahead: the next step's pool is drawn in the shadow of the
// optimizer kernel, one launch and ~13 us of serial latency fewer per training step).
#pragma once
#include "common.h"

#include <rocrand/rocrand_kernel.h>  // device-side Philox4x32-10 (header only): the optional non-parity draw

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
    // rng_kind 1: the pool comes from rocRAND's Philox4x32-10 instead of numpy's MT19937 stream (NOT the reference's
    // negatives: same distribution, other numbers).  Counter based: pool entry p of draw number `fast_draw` is the first
    // accepted output of subsequence fast_draw * P + p -- no generator state on the device, every lane draws on its own
    int rng_kind;
    unsigned long long fast_seed, fast_draw;
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
    if (D.rng_kind == 1 && rng != 0) {  // rocRAND: uniform on [0, rng] by masked rejection, like numpy's randint
        for (int p2 = tid; p2 < P; p2 += NT) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(D.fast_seed, D.fast_draw * (unsigned long long)P + (unsigned long long)p2, 0ull, &st);
            uint32_t v;
            do { v = rocrand(&st) & mask; } while (v > rng);
            pool[p2] = (int64_t)v;
            if (pool2) pool2[p2] = (int64_t)v;
            pool_out_l[p2] = v;
        }
        have = P;
    }
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

// One filter dictionary (negative_sampling.py:7-28) on the device: an open-addressing hash table whose entries
// carry everything a row needs in ONE 32-byte load (no dependent key -> offsets -> flags chain), the concatenated
// sorted true sets, and an entity bitmap for every set of kBitmapMin+ elements (the 3,612-head sets of FB15k-237's
// hub tails made their rows -- and therefore the whole kernel -- 3x slower when streamed element by element).
struct HEntry {
    int64_t key;      // -1 = empty slot
    int64_t off;      // start of the set in `values`
    int32_t len;      // elements in the set
    int32_t flags;    // bit 0: np.in1d takes its sort path for this set; bits 1..: 1 + bitmap index (0 = no bitmap)
    int64_t pad;
};
constexpr int kBitmapMin = 128;

struct Csr {
    HEntry *htab = nullptr;   // capacity = pow2 >= 2 nk
    int64_t *values = nullptr;
    uint32_t *bitmaps = nullptr;  // [n_bitmaps][bm_words]
    uint32_t hmask = 0;
    int bm_words = 0;
    int64_t nk = 0;
};

__host__ __device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t bloom_hash(int32_t v) { return (uint32_t)v * 2654435761u >> 7; }

__device__ __forceinline__ int64_t lower_bound_dev(const int64_t *__restrict__ a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one wave per row, 4 rows per workgroup.  Dynamic LDS: the sorted pool (P2 values + P2 positions, shared by the
// 4 waves) and per wave P kept positions + P ranks + P/32 membership words.
// Membership is searched the cheap way round: the row's true set streams from global memory with coalesced
// loads and each element is binary-searched in the SORTED POOL held in LDS (m log P LDS steps), instead of
// binary-searching global memory for each of the P candidates (P log m dependent global loads).
struct FilterArgs {
    const int64_t *sample;   // [B, 3]
    int B, head_mode;
    int64_t key_stride;
    Csr csr;
    const int64_t *pool;     // [P] the batch's candidate pool and its helper tables (pool_draw_body)
    const uint8_t *lastflag;
    const int32_t *sorted_val, *sorted_pos;
    int K, P, P2, rows_per_wg;
    int64_t *neg;            // [B, K]
    int32_t *posmap;         // [B, K] or null
    uint16_t *cnt;           // [B, P] or null
    int64_t *touched;        // [P + 2B] or null
    int64_t *pool_out;       // null, or the caller's copy of the pool (when it was drawn ahead)
    int32_t *status;
};
inline size_t filter_lds_bytes(int P, int P2, int rw) {
    return sizeof(int32_t) * ((size_t)3 * P2 + (size_t)rw * ((size_t)2 * P + (P + 31) / 32));
}

// one wave per row, rows_per_wg (<= NT / 64) rows per NT-lane workgroup; `block` = index among the filter workgroups;
// lds_i32 = filter_lds_bytes() bytes of dynamic LDS.
template <int NT>
__device__ __forceinline__ void filter_rows_body(const FilterArgs &F, const int block, int32_t *lds_i32) {
    const int64_t *__restrict__ sample = F.sample;
    const int B = F.B, head_mode = F.head_mode, K = F.K, P = F.P, P2 = F.P2, rows_per_wg = F.rows_per_wg;
    const int64_t key_stride = F.key_stride;
    const Csr &csr = F.csr;
    const int64_t *__restrict__ pool = F.pool;
    const uint8_t *__restrict__ lastflag = F.lastflag;
    const int32_t *__restrict__ sorted_val = F.sorted_val, *__restrict__ sorted_pos = F.sorted_pos;
    int64_t *__restrict__ neg = F.neg, *__restrict__ touched = F.touched, *__restrict__ pool_out = F.pool_out;
    int32_t *__restrict__ posmap = F.posmap, *__restrict__ status = F.status;
    uint16_t *__restrict__ cnt = F.cnt;

    int32_t *sval = lds_i32, *spos = lds_i32 + P2;
    if (touched && block == 0)  // id list of the rows a training step reads: pool | heads | tails
        for (int e = threadIdx.x; e < P; e += NT) touched[e] = pool[e];
    if (pool_out && block == 0)  // the pool was drawn ahead of this call: hand the caller its copy
        for (int e = threadIdx.x; e < P; e += NT) pool_out[e] = pool[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int words = (P + 31) / 32;
    const int wslot = wave < rows_per_wg ? wave : 0;  // idle waves alias slot 0 but never touch it
    int32_t *kept = lds_i32 + 2 * P2 + (size_t)wslot * (2 * P + words);  // kept[rho] = position of the rho-th survivor
    int32_t *rank = kept + P;                                            // rank[p] = rho or -1
    uint32_t *member = reinterpret_cast<uint32_t *>(rank + P);           // bit p: pool[p] is in the row's true set
    // Bloom bitmap of the pool's entity ids: 32 bits per pool slot (P2 words), one hash
    uint32_t *bloom = reinterpret_cast<uint32_t *>(lds_i32 + 2 * P2 + (size_t)rows_per_wg * (2 * P + words));
    const uint32_t bloom_mask = (uint32_t)P2 * 32u - 1u;
    for (int e = threadIdx.x; e < P2; e += NT) { sval[e] = sorted_val[e]; spos[e] = sorted_pos[e]; bloom[e] = 0; }
    if (wave < rows_per_wg) for (int w = lane; w < words; w += 64) member[w] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < P2; e += NT) {
        if (spos[e] >= 0) {
            const uint32_t hb = bloom_hash(sval[e]) & bloom_mask;
            atomicOr(&bloom[hb >> 5], 1u << (hb & 31));
        }
    }
    __syncthreads();
    const int i = block * rows_per_wg + wave;
    const bool valid = wave < rows_per_wg && i < B;
    bool found = false;
    int nf = 0;
    if (valid) {
        const int64_t h = sample[3 * (int64_t)i], r = sample[3 * (int64_t)i + 1], t = sample[3 * (int64_t)i + 2];
        if (touched && lane == 0) { touched[P + i] = h; touched[P + B + i] = t; }
        const int64_t key = head_mode ? r * key_stride + t : h * key_stride + r;
        HEntry ent{-1, 0, 0, 0, 0};
        if (csr.nk > 0) {
            uint32_t slot = (uint32_t)mix64((uint64_t)key) & csr.hmask;
            for (;;) {
                ent = csr.htab[slot];
                if (ent.key == key || ent.key < 0) break;
                slot = (slot + 1) & csr.hmask;
            }
        }
        found = ent.key == key;
        if (found) {
            const int m = ent.len;
            const int64_t *rec = csr.values + ent.off;
            const bool sortpath = ent.flags & 1;
            const int bm = (ent.flags >> 1) - 1;
            if (bm >= 0) {  // big set: one independent bitmap probe per pool entry
                const uint32_t *bits = csr.bitmaps + (size_t)bm * csr.bm_words;
                for (int base = 0; base < P; base += 64) {
                    const int p = base + lane;
                    bool mem = false;
                    if (p < P) {
                        const int64_t c = pool[p];
                        mem = (bits[c >> 5] >> (c & 31)) & 1u;
                    }
                    const unsigned long long b = __ballot(mem);
                    if (lane == 0) {
                        member[base >> 5] = (uint32_t)b;
                        if ((base >> 5) + 1 < words) member[(base >> 5) + 1] = (uint32_t)(b >> 32);
                    }
                }
            } else
            // Stream the true set 8 elements per lane at a time (independent coalesced loads in flight together),
            // probe a Bloom bitmap of the pool first: almost every element misses and costs one LDS read; the
            // rare hit is confirmed (and its positions found) by binary search in the sorted pool.
            for (int e0 = 0; e0 < m; e0 += 64 * 8) {
                int64_t v64[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * 64 + lane;
                    v64[u] = e < m ? rec[e] : -1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (v64[u] < 0 || v64[u] >= INT32_MAX) continue;
                    const int32_t v = (int32_t)v64[u];
                    const uint32_t hb = bloom_hash(v) & bloom_mask;
                    if (!((bloom[hb >> 5] >> (hb & 31)) & 1u)) continue;
                    int lo = 0, hi = P2;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (sval[mid] < v) lo = mid + 1; else hi = mid;
                    }
                    for (; lo < P2 && sval[lo] == v; ++lo) {
                        const int p = spos[lo];
                        atomicOr(&member[p >> 5], 1u << (p & 31));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < P; base += 64) {
                const int p = base + lane;
                bool keep = false;
                if (p < P) {
                    const bool mem = (member[p >> 5] >> (p & 31)) & 1u;
                    keep = !mem && (!sortpath || lastflag[p] != 0);
                }
                const unsigned long long b = __ballot(keep);
                const int rho = nf + __popcll(b & ((1ull << lane) - 1ull));
                if (keep) kept[rho] = p;
                if (p < P) rank[p] = keep ? rho : -1;
                nf += __popcll(b);
            }
        }
    }
    __syncthreads();  // kept[] / rank[] visible to every lane of the wave that wrote them
    if (!valid) return;
    if (!found || nf == 0) {
        if (lane == 0) {
            atomicCAS(&status[0], 0, found ? (int)MKB_ERR_EMPTY : (int)MKB_ERR_KEY);
            atomicMin(&status[1], i);
        }
        for (int j = lane; j < K; j += 64) {
            neg[(int64_t)i * K + j] = 0;
            if (posmap) posmap[(int64_t)i * K + j] = 0;
        }
        if (cnt) for (int p = lane; p < P; p += 64) cnt[(int64_t)i * P + p] = 0;
        return;
    }
    for (int j = lane; j < K; j += 64) {  // cyclic fill: concat(f, f, ...)[:K]   (negative_sampling.py:176-195)
        const int pp = kept[j % nf];
        neg[(int64_t)i * K + j] = pool[pp];
        if (posmap) posmap[(int64_t)i * K + j] = pp;
    }
    if (cnt) {  // multiplicity of pool position p among the K slots of this row
        for (int p = lane; p < P; p += 64) {
            const int rho = rank[p];
            cnt[(int64_t)i * P + p] = (rho >= 0 && rho < K) ? (uint16_t)((K - 1 - rho) / nf + 1) : (uint16_t)0;
        }
    }
}

}  // namespace mkb

// sampler.hip: hand the next pool draw of `s` to another launch (fills *D, returns the dynamic LDS it needs); the next
// mkb_sampler_generate then skips its own draw kernel.  false: nothing to piggyback (a drawn pool is already waiting).
struct mkb_sampler;
namespace mkb {
bool sampler_draw_ahead(mkb_sampler *s, DrawArgs *D, size_t *lds_bytes);
// The whole of mkb_sampler_generate for another launch to carry: *F = the row filter of THIS batch (its pool must have
// been drawn ahead; if not, the draw kernel is launched on `st` first), *D = the draw of the NEXT pool into the other
// buffer.  *pool_ids = this batch's pool on the device (valid until the generate after next).
int sampler_ride(mkb_sampler *s, const int64_t *sample, int64_t B, int mode, int64_t *neg, int64_t *pool, int32_t *pos,
                 uint16_t *cnt, int64_t *touched, FilterArgs *F, DrawArgs *D, const int64_t **pool_ids, size_t *lds_bytes,
                 hipStream_t st, int carrier_lanes = 1024);  // carrier_lanes: workgroup size of the launch that carries the rows
}

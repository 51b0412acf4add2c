// On-device filtered uniform negative sampler, bit-exact with the reference's numpy path.
//
// Replaces sampling.NegativeSampling.generate (sampling/negative_sampling.py:158-201):
//   pool = RandomState.randint(n_entity, size=2K)            ONE draw per batch (:166)
//   row i: keep = np.in1d(pool, true_set_i, assume_unique=True, invert=True)   (:153-156)
//          out_i = cyclic(pool[keep])[:K]                                        (:176-199)
// numpy pieces restated (numpy 2.2.6): legacy MT19937 (init_genrand seeding, 624-word block regeneration,
// tempering), masked-rejection bounded draw (one 32-bit output per trial), and the three np.in1d branches
// (table / loop = exact membership; sort = membership AND only the LAST occurrence of a duplicated
// candidate survives).  The branch taken depends only on (len, min, max) of the row's true set and on 2K,
// so it is decided once per key on the host at create time.
//
// Kernel 1 (ONE workgroup, 1024 lanes): the MT19937 stream is serial in its *consumption* (rejections shift
//   everything after them) but a 624-word block can be regenerated in three data-parallel phases
//   (i<227 reads only old words; 227<=i<454 reads new words of phase 1; i>=454 reads new words of phase 2),
//   tempered and masked in parallel, and the accepted draws compacted with a ballot/prefix scan; the
//   stream position advances to just after the 2K-th accept, exactly like the serial loop.
// Kernel 2 (one wave per batch row): binary-search the row's true set for each of the 2K pool entries,
//   ballot-compact the kept positions into LDS, emit the K negatives cyclically plus the pooled-path
//   side outputs (position map, per-position multiplicity).
// No host round trip: errors (unseen key -> KeyError in the reference, empty filter -> infinite loop in the
// reference) are recorded in a device status word that the host reads lazily.
#include "common.h"
#include "sampler_draw.h"

#include <math.h>
#include <vector>

namespace mkb {

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

}  // namespace mkb

struct mkb_sampler {
    int64_t n_entity, n_relation, K;
    uint32_t *mt;      // device [624]
    int32_t *mtpos;    // device [1]
    int32_t *status;   // device [2]: code, row
    int64_t *pool;     // device [2K] (internal copy when the caller passes none)
    uint8_t *lastflag; // device [2K]
    int32_t *sorted_val, *sorted_pos;  // device [P2]: the pool sorted by (entity, position), P2 = pow2 >= 2K
    uint32_t *mt_prev; // device [625]: generator state before a pool drawn ahead (sampler_draw_ahead)
    bool drawn_ahead;  // the next generate's pool is already in pool / lastflag / sorted_*
    mkb::Csr head, tail;
};

namespace mkb {

__global__ __launch_bounds__(1024) void pool_draw_kernel(DrawArgs D) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skey[];  // [P2] sort keys, then [P] drawn values
    pool_draw_body<1024>(D, skey);
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
__global__ __launch_bounds__(256) void filter_rows_kernel(const int64_t *__restrict__ sample, int B, int head_mode,
                                                          int64_t key_stride, Csr csr,
                                                          const int64_t *__restrict__ pool,
                                                          const uint8_t *__restrict__ lastflag,
                                                          const int32_t *__restrict__ sorted_val,
                                                          const int32_t *__restrict__ sorted_pos, int K, int P, int P2,
                                                          int rows_per_wg, int64_t *__restrict__ neg, int32_t *__restrict__ posmap,
                                                          uint16_t *__restrict__ cnt, int64_t *__restrict__ touched,
                                                          int64_t *__restrict__ pool_out, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds_i32[];
    int32_t *sval = lds_i32, *spos = lds_i32 + P2;
    if (touched && blockIdx.x == 0)  // id list of the rows a training step reads: pool | heads | tails
        for (int e = threadIdx.x; e < P; e += 256) touched[e] = pool[e];
    if (pool_out && blockIdx.x == 0)  // the pool was drawn ahead of this call: hand the caller its copy
        for (int e = threadIdx.x; e < P; e += 256) pool_out[e] = pool[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int words = (P + 31) / 32;
    const int wslot = wave < rows_per_wg ? wave : 0;  // idle waves alias slot 0 but never touch it
    int32_t *kept = lds_i32 + 2 * P2 + (size_t)wslot * (2 * P + words);  // kept[rho] = position of the rho-th survivor
    int32_t *rank = kept + P;                                            // rank[p] = rho or -1
    uint32_t *member = reinterpret_cast<uint32_t *>(rank + P);           // bit p: pool[p] is in the row's true set
    // Bloom bitmap of the pool's entity ids: 32 bits per pool slot (P2 words), one hash
    uint32_t *bloom = reinterpret_cast<uint32_t *>(lds_i32 + 2 * P2 + (size_t)rows_per_wg * (2 * P + words));
    const uint32_t bloom_mask = (uint32_t)P2 * 32u - 1u;
    for (int e = threadIdx.x; e < P2; e += 256) { sval[e] = sorted_val[e]; spos[e] = sorted_pos[e]; bloom[e] = 0; }
    if (wave < rows_per_wg) for (int w = lane; w < words; w += 64) member[w] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < P2; e += 256) {
        if (spos[e] >= 0) {
            const uint32_t hb = bloom_hash(sval[e]) & bloom_mask;
            atomicOr(&bloom[hb >> 5], 1u << (hb & 31));
        }
    }
    __syncthreads();
    const int i = blockIdx.x * rows_per_wg + wave;
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

static int upload_csr(Csr &c, const int64_t *keys, int64_t nk, const int64_t *offsets, const int64_t *values, int64_t P,
                      int64_t n_entity, hipStream_t st) {
    MKB_REQUIRE(nk >= 0 && nk < (1ll << 30) && (nk == 0 || (keys && offsets && values)), "bad CSR");
    c.nk = nk;
    const int64_t nv = nk ? offsets[nk] : 0;
    const double loop_thr = 10.0 * pow((double)P, 0.145);  // numpy: len(ar2) < 10 * len(ar1) ** 0.145
    uint32_t cap = 2;
    while ((int64_t)cap < 2 * nk) cap <<= 1;
    std::vector<HEntry> htab(cap, HEntry{-1, 0, 0, 0, 0});
    c.hmask = cap - 1;
    c.bm_words = (int)((n_entity + 31) / 32);
    std::vector<uint32_t> bitmaps;
    int n_bm = 0;
    for (int64_t k = 0; k < nk; ++k) {
        const int64_t m = offsets[k + 1] - offsets[k];
        MKB_REQUIRE(m > 0 && m < INT32_MAX, "bad true-set size for key %lld", (long long)k);
        const int64_t range = values[offsets[k + 1] - 1] - values[offsets[k]];
        const bool table = range <= 6 * (P + m);   // numpy >= 1.24 _in1d: table / loop / sort path selection
        const bool loop = (double)m < loop_thr;
        int32_t flags = (!table && !loop) ? 1 : 0;
        if (m >= kBitmapMin && (int64_t)(n_bm + 1) * c.bm_words * 4 <= (256ll << 20)) {
            bitmaps.resize((size_t)(n_bm + 1) * c.bm_words, 0u);
            uint32_t *b = bitmaps.data() + (size_t)n_bm * c.bm_words;
            for (int64_t e = offsets[k]; e < offsets[k + 1]; ++e) b[values[e] >> 5] |= 1u << (values[e] & 31);
            flags |= (n_bm + 1) << 1;
            ++n_bm;
        }
        uint32_t slot = (uint32_t)mix64((uint64_t)keys[k]) & c.hmask;
        while (htab[slot].key >= 0) slot = (slot + 1) & c.hmask;
        htab[slot] = HEntry{keys[k], offsets[k], (int32_t)m, flags, 0};
    }
    MKB_CHECK_HIP(hipMalloc(&c.htab, sizeof(HEntry) * cap));
    MKB_CHECK_HIP(hipMalloc(&c.values, sizeof(int64_t) * (size_t)(nv ? nv : 1)));
    MKB_CHECK_HIP(hipMalloc(&c.bitmaps, sizeof(uint32_t) * (bitmaps.empty() ? 1 : bitmaps.size())));
    MKB_CHECK_HIP(hipMemcpyAsync(c.htab, htab.data(), sizeof(HEntry) * cap, hipMemcpyHostToDevice, st));
    if (nv) MKB_CHECK_HIP(hipMemcpyAsync(c.values, values, sizeof(int64_t) * (size_t)nv, hipMemcpyHostToDevice, st));
    if (!bitmaps.empty())
        MKB_CHECK_HIP(hipMemcpyAsync(c.bitmaps, bitmaps.data(), sizeof(uint32_t) * bitmaps.size(), hipMemcpyHostToDevice, st));
    MKB_CHECK_HIP(hipStreamSynchronize(st));  // host vectors die at return
    return MKB_OK;
}

static void free_csr(Csr &c) {
    (void)hipFree(c.htab); (void)hipFree(c.values); (void)hipFree(c.bitmaps);
    c = Csr();
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_sampler_create(mkb_sampler_t **out, int64_t n_entity, int64_t n_relation, int64_t K, uint32_t seed,
                                  const int64_t *head_keys_host, int64_t n_head_keys, const int64_t *head_offsets_host,
                                  const int64_t *head_values_host, const int64_t *tail_keys_host, int64_t n_tail_keys,
                                  const int64_t *tail_offsets_host, const int64_t *tail_values_host, void *stream) {
    MKB_REQUIRE(out != nullptr, "out is null");
    MKB_REQUIRE(n_entity > 0 && n_entity < INT32_MAX && n_relation > 0, "bad n_entity / n_relation");
    MKB_REQUIRE(K > 0 && K <= 1024, "size must be in [1, 1024] for the device sampler");
    hipStream_t st = (hipStream_t)stream;
    mkb_sampler *s = new mkb_sampler();
    s->n_entity = n_entity; s->n_relation = n_relation; s->K = K;
    s->mt = nullptr; s->mtpos = nullptr; s->status = nullptr; s->pool = nullptr; s->lastflag = nullptr;
    s->mt_prev = nullptr; s->drawn_ahead = false;
    s->sorted_val = nullptr; s->sorted_pos = nullptr;
    int P2 = 2;
    while (P2 < 2 * K) P2 <<= 1;
    auto fail = [&](int rc) { mkb_sampler_destroy(s); return rc; };
    if (hipMalloc(&s->mt, sizeof(uint32_t) * MT_N) != hipSuccess || hipMalloc(&s->mtpos, sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&s->status, 2 * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&s->mt_prev, sizeof(uint32_t) * (MT_N + 1)) != hipSuccess ||
        hipMalloc(&s->pool, sizeof(int64_t) * (size_t)(2 * K)) != hipSuccess ||
        hipMalloc(&s->lastflag, (size_t)(2 * K)) != hipSuccess ||
        hipMalloc(&s->sorted_val, sizeof(int32_t) * (size_t)P2) != hipSuccess ||
        hipMalloc(&s->sorted_pos, sizeof(int32_t) * (size_t)P2) != hipSuccess)
        return fail(set_error(MKB_ERR_HIP, "hipMalloc failed in mkb_sampler_create"));
    uint32_t key[MT_N];
    uint32_t sd = seed;  // numpy mt19937_seed == init_genrand
    for (int i = 0; i < MT_N; ++i) {
        key[i] = sd;
        sd = 1812433253u * (sd ^ (sd >> 30)) + (uint32_t)i + 1u;
    }
    if (int rc = mkb_sampler_set_state(s, key, MT_N, stream)) return fail(rc);
    if (int rc = upload_csr(s->head, head_keys_host, n_head_keys, head_offsets_host, head_values_host, 2 * K, n_entity, st)) return fail(rc);
    if (int rc = upload_csr(s->tail, tail_keys_host, n_tail_keys, tail_offsets_host, tail_values_host, 2 * K, n_entity, st)) return fail(rc);
    *out = s;
    return MKB_OK;
}

extern "C" void mkb_sampler_destroy(mkb_sampler_t *s) {
    if (!s) return;
    (void)hipFree(s->mt); (void)hipFree(s->mtpos); (void)hipFree(s->status); (void)hipFree(s->pool); (void)hipFree(s->lastflag);
    (void)hipFree(s->sorted_val); (void)hipFree(s->sorted_pos); (void)hipFree(s->mt_prev);
    free_csr(s->head);
    free_csr(s->tail);
    delete s;
}

extern "C" int mkb_sampler_set_state(mkb_sampler_t *s, const uint32_t *key624_host, int32_t pos, void *stream) {
    MKB_REQUIRE(s && key624_host && pos >= 0 && pos <= MT_N, "bad state");
    hipStream_t st = (hipStream_t)stream;
    s->drawn_ahead = false;  // a pool drawn ahead from the old state is discarded
    const int32_t zero[2] = {0, INT32_MAX};
    MKB_CHECK_HIP(hipMemcpyAsync(s->mt, key624_host, sizeof(uint32_t) * MT_N, hipMemcpyHostToDevice, st));
    MKB_CHECK_HIP(hipMemcpyAsync(s->mtpos, &pos, sizeof(int32_t), hipMemcpyHostToDevice, st));
    MKB_CHECK_HIP(hipMemcpyAsync(s->status, zero, sizeof(zero), hipMemcpyHostToDevice, st));
    MKB_CHECK_HIP(hipStreamSynchronize(st));
    return MKB_OK;
}

extern "C" int mkb_sampler_get_state(mkb_sampler_t *s, uint32_t *key624_host, int32_t *pos_host, void *stream) {
    MKB_REQUIRE(s && key624_host && pos_host, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (s->drawn_ahead) {  // report the state the NEXT generate() logically starts from: before the pool drawn ahead
        uint32_t p = 0;
        MKB_CHECK_HIP(hipMemcpyAsync(key624_host, s->mt_prev, sizeof(uint32_t) * MT_N, hipMemcpyDeviceToHost, st));
        MKB_CHECK_HIP(hipMemcpyAsync(&p, s->mt_prev + MT_N, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        MKB_CHECK_HIP(hipStreamSynchronize(st));
        *pos_host = (int32_t)p;
        return MKB_OK;
    }
    MKB_CHECK_HIP(hipMemcpyAsync(key624_host, s->mt, sizeof(uint32_t) * MT_N, hipMemcpyDeviceToHost, st));
    MKB_CHECK_HIP(hipMemcpyAsync(pos_host, s->mtpos, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MKB_CHECK_HIP(hipStreamSynchronize(st));
    return MKB_OK;
}

extern "C" int mkb_sampler_status(mkb_sampler_t *s, void *stream) {
    MKB_REQUIRE(s != nullptr, "sampler is null");
    hipStream_t st = (hipStream_t)stream;
    int32_t h[2] = {0, 0};
    MKB_CHECK_HIP(hipMemcpyAsync(h, s->status, sizeof(h), hipMemcpyDeviceToHost, st));
    MKB_CHECK_HIP(hipStreamSynchronize(st));
    if (h[0] == MKB_ERR_KEY)
        return set_error(MKB_ERR_KEY, "row %d: its (relation, tail) / (head, relation) pair is not in the training triples", h[1]);
    if (h[0] == MKB_ERR_EMPTY)
        return set_error(MKB_ERR_EMPTY, "row %d: the filter removed the whole candidate pool (the reference never returns here)", h[1]);
    return MKB_OK;
}

bool mkb::sampler_draw_ahead(mkb_sampler *s, mkb::DrawArgs *D, size_t *lds_bytes) {
    if (!s || s->drawn_ahead) return false;
    const int P = (int)(2 * s->K);
    int P2 = 2;
    while (P2 < P) P2 <<= 1;
    *D = mkb::DrawArgs{s->mt, s->mtpos, s->mt_prev, (uint32_t)(s->n_entity - 1), P, P2, s->pool, nullptr, s->lastflag,
                       s->sorted_val, s->sorted_pos};
    *lds_bytes = mkb::draw_lds_bytes(P, P2);
    s->drawn_ahead = true;
    return true;
}

extern "C" int mkb_sampler_generate(mkb_sampler_t *s, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                                    int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream) {
    MKB_REQUIRE(s && sample && neg, "null pointer");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "generate needs head-batch or tail-batch");
    MKB_REQUIRE(B >= 0 && B <= INT32_MAX, "bad B");
    hipStream_t st = (hipStream_t)stream;
    const int P = (int)(2 * s->K);
    int P2 = 2;
    while (P2 < P) P2 <<= 1;
    ProfScope ps(MKB_PROF_SAMPLER, st);
    const bool ahead = s->drawn_ahead;  // the pool was drawn inside an earlier launch (sampler_draw_ahead)
    s->drawn_ahead = false;
    if (!ahead) {
        DrawArgs D{s->mt, s->mtpos, nullptr, (uint32_t)(s->n_entity - 1), P, P2, s->pool, pool, s->lastflag, s->sorted_val,
                   s->sorted_pos};
        hipLaunchKernelGGL(pool_draw_kernel, dim3(1), dim3(1024), draw_lds_bytes(P, P2), st, D);
    }
    MKB_LAUNCH_CHECK();
    if (B == 0) {
        if (ahead && pool) MKB_CHECK_HIP(hipMemcpyAsync(pool, s->pool, sizeof(int64_t) * (size_t)P, hipMemcpyDeviceToDevice, st));
        return MKB_OK;
    }
    const bool head = mode == MKB_MODE_HEAD;
    const Csr &c = head ? s->head : s->tail;
    const int64_t stride = head ? s->n_entity : s->n_relation;
    const int rw = P <= 1024 ? 4 : 1;  // rows (waves) per workgroup: keeps the LDS request under 64 KB up to P = 2048
    const size_t lds = sizeof(int32_t) * ((size_t)3 * P2 + (size_t)rw * ((size_t)2 * P + (P + 31) / 32));
    hipLaunchKernelGGL(filter_rows_kernel, dim3((unsigned)((B + rw - 1) / rw)), dim3(256), lds, st, sample, (int)B,
                       head ? 1 : 0, stride, c, s->pool, s->lastflag, s->sorted_val, s->sorted_pos, (int)s->K, P, P2, rw,
                       neg, pos, cnt, touched, ahead ? pool : nullptr, s->status);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

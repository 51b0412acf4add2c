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

#include <math.h>
#include <vector>

namespace mkb {

constexpr int MT_N = 624, MT_M = 397;

struct Csr {
    int64_t *keys = nullptr, *offsets = nullptr, *values = nullptr;
    uint8_t *sortflag = nullptr;
    int64_t nk = 0;
};

}  // namespace mkb

struct mkb_sampler {
    int64_t n_entity, n_relation, K;
    uint32_t *mt;      // device [624]
    int32_t *mtpos;    // device [1]
    int32_t *status;   // device [2]: code, row
    int64_t *pool;     // device [2K] (internal copy when the caller passes none)
    uint8_t *lastflag; // device [2K]
    mkb::Csr head, tail;
};

namespace mkb {

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

// exclusive prefix sum of a 0/1 flag over a 1024-thread block; returns rank, *total = block total
__device__ __forceinline__ int block_scan_flag(bool flag, int *wave_tot, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wave_tot[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int c = wave_tot[w];
        if (w < wave) base += c;
        tot += c;
    }
    *total = tot;
    return base + in_wave;
}

__global__ __launch_bounds__(1024) void pool_draw_kernel(uint32_t *__restrict__ mt_g, int32_t *__restrict__ pos_g,
                                                         uint32_t rng, int P, int64_t *__restrict__ pool,
                                                         int64_t *__restrict__ pool2, uint8_t *__restrict__ lastflag) {
    __shared__ uint32_t mt[MT_N];
    __shared__ int wave_tot[16];
    __shared__ int s_newpos;
    const int tid = threadIdx.x;
    if (tid < MT_N) mt[tid] = mt_g[tid];
    int pos = pos_g[0];
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    __syncthreads();
    int have = 0;
    if (rng == 0) {  // randint(1): no stream consumption
        for (int p = tid; p < P; p += 1024) { pool[p] = 0; if (pool2) pool2[p] = 0; }
        have = P;
    }
    while (have < P) {
        if (pos == MT_N) {  // regenerate the block: three parallel phases + the last word
            uint32_t nv = 0;
            if (tid < 227) nv = mt_twist(mt[tid], mt[tid + 1], mt[tid + MT_M]);
            __syncthreads();
            if (tid < 227) mt[tid] = nv;
            __syncthreads();
            if (tid >= 227 && tid < 454) nv = mt_twist(mt[tid], mt[tid + 1], mt[tid - 227]);
            __syncthreads();
            if (tid >= 227 && tid < 454) mt[tid] = nv;
            __syncthreads();
            if (tid >= 454 && tid < 623) nv = mt_twist(mt[tid], mt[tid + 1], mt[tid - 227]);
            __syncthreads();
            if (tid >= 454 && tid < 623) mt[tid] = nv;
            __syncthreads();
            if (tid == 623) mt[623] = mt_twist(mt[623], mt[0], mt[396]);
            __syncthreads();
            pos = 0;
        }
        const int avail = MT_N - pos;
        uint32_t v = 0;
        bool acc = false;
        if (tid < avail) {
            v = mt_temper(mt[pos + tid]) & mask;
            acc = v <= rng;
        }
        int total;
        const int rank = block_scan_flag(acc, wave_tot, &total);
        const int need = P - have;
        if (acc && rank < need) {
            pool[have + rank] = (int64_t)v;
            if (pool2) pool2[have + rank] = (int64_t)v;
        }
        if (tid == 0) s_newpos = MT_N;
        __syncthreads();
        if (acc && rank == need - 1) s_newpos = pos + tid + 1;  // word that produced the last needed draw
        __syncthreads();
        pos = s_newpos;
        have += (total < need) ? total : need;
        __syncthreads();
    }
    if (tid < MT_N) mt_g[tid] = mt[tid];
    if (tid == 0) pos_g[0] = pos;
    // lastflag[p] = no later pool position holds the same entity (the np.in1d sort path keeps only those)
    __threadfence_block();
    __syncthreads();
    for (int p = tid; p < P; p += 1024) {
        const int64_t c = pool[p];
        uint8_t last = 1;
        for (int q = p + 1; q < P; ++q)
            if (pool[q] == c) { last = 0; break; }
        lastflag[p] = last;
    }
}

__device__ __forceinline__ int64_t lower_bound_dev(const int64_t *__restrict__ a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one wave per row, 4 rows per workgroup; dynamic LDS: per wave P int32 kept positions + P int32 ranks
__global__ __launch_bounds__(256) void filter_rows_kernel(const int64_t *__restrict__ sample, int B, int head_mode,
                                                          int64_t key_stride, const int64_t *__restrict__ keys, int64_t nk,
                                                          const int64_t *__restrict__ offsets,
                                                          const int64_t *__restrict__ values,
                                                          const uint8_t *__restrict__ sortflag,
                                                          const int64_t *__restrict__ pool,
                                                          const uint8_t *__restrict__ lastflag, int K, int P,
                                                          int64_t *__restrict__ neg, int32_t *__restrict__ posmap,
                                                          uint16_t *__restrict__ cnt, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds_i32[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    const bool valid = i < B;
    int32_t *kept = lds_i32 + (size_t)wave * 2 * P;  // kept[rho] = pool position of the rho-th surviving candidate
    int32_t *rank = kept + P;                        // rank[p]   = rho, or -1 when position p is filtered out
    bool found = false;
    int nf = 0;
    if (valid) {
        const int64_t h = sample[3 * (int64_t)i], r = sample[3 * (int64_t)i + 1], t = sample[3 * (int64_t)i + 2];
        const int64_t key = head_mode ? r * key_stride + t : h * key_stride + r;
        const int64_t ki = lower_bound_dev(keys, nk, key);
        found = ki < nk && keys[ki] == key;
        if (found) {
            const int64_t off = offsets[ki];
            const int64_t m = offsets[ki + 1] - off;
            const int64_t *rec = values + off;
            const bool sortpath = sortflag[ki] != 0;
            for (int base = 0; base < P; base += 64) {
                const int p = base + lane;
                bool keep = false;
                if (p < P) {
                    const int64_t c = pool[p];
                    const int64_t j = lower_bound_dev(rec, m, c);
                    const bool member = j < m && rec[j] == c;
                    keep = !member && (!sortpath || lastflag[p] != 0);
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
                      hipStream_t st) {
    MKB_REQUIRE(nk >= 0 && (nk == 0 || (keys && offsets && values)), "bad CSR");
    c.nk = nk;
    const int64_t nv = nk ? offsets[nk] : 0;
    std::vector<uint8_t> flag((size_t)(nk ? nk : 1), 0);
    const double loop_thr = 10.0 * pow((double)P, 0.145);  // numpy: len(ar2) < 10 * len(ar1) ** 0.145
    for (int64_t k = 0; k < nk; ++k) {
        const int64_t m = offsets[k + 1] - offsets[k];
        MKB_REQUIRE(m > 0, "empty true-set for key %lld", (long long)k);
        const int64_t range = values[offsets[k + 1] - 1] - values[offsets[k]];
        const bool table = range <= 6 * (P + m);
        const bool loop = (double)m < loop_thr;
        flag[(size_t)k] = (!table && !loop) ? 1 : 0;
    }
    MKB_CHECK_HIP(hipMalloc(&c.keys, sizeof(int64_t) * (size_t)(nk ? nk : 1)));
    MKB_CHECK_HIP(hipMalloc(&c.offsets, sizeof(int64_t) * (size_t)(nk + 1)));
    MKB_CHECK_HIP(hipMalloc(&c.values, sizeof(int64_t) * (size_t)(nv ? nv : 1)));
    MKB_CHECK_HIP(hipMalloc(&c.sortflag, (size_t)(nk ? nk : 1)));
    if (nk) {
        MKB_CHECK_HIP(hipMemcpyAsync(c.keys, keys, sizeof(int64_t) * (size_t)nk, hipMemcpyHostToDevice, st));
        MKB_CHECK_HIP(hipMemcpyAsync(c.offsets, offsets, sizeof(int64_t) * (size_t)(nk + 1), hipMemcpyHostToDevice, st));
        MKB_CHECK_HIP(hipMemcpyAsync(c.values, values, sizeof(int64_t) * (size_t)nv, hipMemcpyHostToDevice, st));
        MKB_CHECK_HIP(hipMemcpyAsync(c.sortflag, flag.data(), (size_t)nk, hipMemcpyHostToDevice, st));
    }
    MKB_CHECK_HIP(hipStreamSynchronize(st));  // host vectors die at return
    return MKB_OK;
}

static void free_csr(Csr &c) {
    (void)hipFree(c.keys); (void)hipFree(c.offsets); (void)hipFree(c.values); (void)hipFree(c.sortflag);
    c = Csr();
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_sampler_create(mkb_sampler_t **out, int64_t n_entity, int64_t n_relation, int64_t K, uint32_t seed,
                                  const int64_t *head_keys_host, int64_t n_head_keys, const int64_t *head_offsets_host,
                                  const int64_t *head_values_host, const int64_t *tail_keys_host, int64_t n_tail_keys,
                                  const int64_t *tail_offsets_host, const int64_t *tail_values_host, void *stream) {
    MKB_REQUIRE(out != nullptr, "out is null");
    MKB_REQUIRE(n_entity > 0 && n_entity <= 0xFFFFFFFFll && n_relation > 0, "bad n_entity / n_relation");
    MKB_REQUIRE(K > 0 && 2 * K <= 8192, "size must be in [1, 4096]");
    hipStream_t st = (hipStream_t)stream;
    mkb_sampler *s = new mkb_sampler();
    s->n_entity = n_entity; s->n_relation = n_relation; s->K = K;
    s->mt = nullptr; s->mtpos = nullptr; s->status = nullptr; s->pool = nullptr; s->lastflag = nullptr;
    auto fail = [&](int rc) { mkb_sampler_destroy(s); return rc; };
    if (hipMalloc(&s->mt, sizeof(uint32_t) * MT_N) != hipSuccess || hipMalloc(&s->mtpos, sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&s->status, 2 * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&s->pool, sizeof(int64_t) * (size_t)(2 * K)) != hipSuccess ||
        hipMalloc(&s->lastflag, (size_t)(2 * K)) != hipSuccess)
        return fail(set_error(MKB_ERR_HIP, "hipMalloc failed in mkb_sampler_create"));
    uint32_t key[MT_N];
    uint32_t sd = seed;  // numpy mt19937_seed == init_genrand
    for (int i = 0; i < MT_N; ++i) {
        key[i] = sd;
        sd = 1812433253u * (sd ^ (sd >> 30)) + (uint32_t)i + 1u;
    }
    if (int rc = mkb_sampler_set_state(s, key, MT_N, stream)) return fail(rc);
    if (int rc = upload_csr(s->head, head_keys_host, n_head_keys, head_offsets_host, head_values_host, 2 * K, st)) return fail(rc);
    if (int rc = upload_csr(s->tail, tail_keys_host, n_tail_keys, tail_offsets_host, tail_values_host, 2 * K, st)) return fail(rc);
    *out = s;
    return MKB_OK;
}

extern "C" void mkb_sampler_destroy(mkb_sampler_t *s) {
    if (!s) return;
    (void)hipFree(s->mt); (void)hipFree(s->mtpos); (void)hipFree(s->status); (void)hipFree(s->pool); (void)hipFree(s->lastflag);
    free_csr(s->head);
    free_csr(s->tail);
    delete s;
}

extern "C" int mkb_sampler_set_state(mkb_sampler_t *s, const uint32_t *key624_host, int32_t pos, void *stream) {
    MKB_REQUIRE(s && key624_host && pos >= 0 && pos <= MT_N, "bad state");
    hipStream_t st = (hipStream_t)stream;
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

extern "C" int mkb_sampler_generate(mkb_sampler_t *s, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                                    int64_t *pool, int32_t *pos, uint16_t *cnt, void *stream) {
    MKB_REQUIRE(s && sample && neg, "null pointer");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "generate needs head-batch or tail-batch");
    MKB_REQUIRE(B >= 0 && B <= INT32_MAX, "bad B");
    hipStream_t st = (hipStream_t)stream;
    const int P = (int)(2 * s->K);
    ProfScope ps(MKB_PROF_SAMPLER, st);
    hipLaunchKernelGGL(pool_draw_kernel, dim3(1), dim3(1024), 0, st, s->mt, s->mtpos, (uint32_t)(s->n_entity - 1), P,
                       s->pool, pool, s->lastflag);
    MKB_LAUNCH_CHECK();
    if (B == 0) return MKB_OK;
    const bool head = mode == MKB_MODE_HEAD;
    const Csr &c = head ? s->head : s->tail;
    const int64_t stride = head ? s->n_entity : s->n_relation;
    hipLaunchKernelGGL(filter_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), (size_t)8 * P * sizeof(int32_t), st,
                       sample, (int)B, head ? 1 : 0, stride, c.keys, c.nk, c.offsets, c.values, c.sortflag, s->pool,
                       s->lastflag, (int)s->K, P, neg, pos, cnt, s->status);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

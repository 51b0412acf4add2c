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

struct mkb_sampler {
    int64_t n_entity, n_relation, K;
    uint32_t *mt;      // device [624]
    int32_t *mtpos;    // device [1]
    int32_t *status;   // device [2]: code, row
    // Pool state, double buffered: buffer `cur` holds the pool of the batch being filtered, the other one receives a pool
    // drawn ahead of the next generate (so that the draw can share a launch with the current batch's filter).
    int64_t *pool;     // device [2][2K]
    uint8_t *lastflag; // device [2][2K]
    int32_t *sorted_val, *sorted_pos;  // device [2][P2]: the pool sorted by (entity, position), P2 = pow2 >= 2K
    uint32_t *mt_prev; // device [625]: generator state before a pool drawn ahead (sampler_draw_ahead)
    int cur;
    bool drawn_ahead;  // the next generate's pool is already in buffer cur ^ 1
    int rng_kind = 0;                              // 0: numpy MT19937 (bit-exact with the reference); 1: rocRAND Philox4x32-10
    unsigned long long fast_seed = 0, fast_draw = 0;  // rocRAND mode: seed and the number of pools drawn so far
    mkb::Csr head, tail;
    int P() const { return (int)(2 * K); }
    int P2() const { int p2 = 2; while (p2 < P()) p2 <<= 1; return p2; }
    mkb::DrawArgs draw_args(int buf, int64_t *pool_out, bool save_prev) {
        mkb::DrawArgs D{mt, mtpos, save_prev ? mt_prev : nullptr, (uint32_t)(n_entity - 1), P(), P2(),
                        pool + (size_t)buf * P(), pool_out, lastflag + (size_t)buf * P(),
                        sorted_val + (size_t)buf * P2(), sorted_pos + (size_t)buf * P2(), rng_kind, fast_seed, fast_draw};
        if (rng_kind == 1) ++fast_draw;  // (every DrawArgs built is one pool drawn: the launch is enqueued by the caller)
        return D;
    }
};

namespace mkb {

__global__ __launch_bounds__(1024) void pool_draw_kernel(DrawArgs D) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skey[];  // [P2] sort keys, then [P] drawn values
    pool_draw_body<1024>(D, skey);
}

__global__ __launch_bounds__(256) void filter_rows_kernel(FilterArgs F) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds_i32[];
    filter_rows_body<256>(F, (int)blockIdx.x, lds_i32);
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
    s->mt_prev = nullptr; s->drawn_ahead = false; s->cur = 0;
    s->sorted_val = nullptr; s->sorted_pos = nullptr;
    int P2 = 2;
    while (P2 < 2 * K) P2 <<= 1;
    auto fail = [&](int rc) { mkb_sampler_destroy(s); return rc; };
    if (hipMalloc(&s->mt, sizeof(uint32_t) * MT_N) != hipSuccess || hipMalloc(&s->mtpos, sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&s->status, 2 * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&s->mt_prev, sizeof(uint32_t) * (MT_N + 1)) != hipSuccess ||
        hipMalloc(&s->pool, 2 * sizeof(int64_t) * (size_t)(2 * K)) != hipSuccess ||
        hipMalloc(&s->lastflag, 2 * (size_t)(2 * K)) != hipSuccess ||
        hipMalloc(&s->sorted_val, 2 * sizeof(int32_t) * (size_t)P2) != hipSuccess ||
        hipMalloc(&s->sorted_pos, 2 * sizeof(int32_t) * (size_t)P2) != hipSuccess)
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

// Optional non-parity draw (BASELINE north_star: "a rocRAND + HIP reject kernel"): kind 1 = rocRAND's Philox4x32-10, counter
// based (no device state: `draws` pools have been drawn so far -- pass 0 for a fresh sampler, or the value of
// mkb_sampler_get_rng to resume).  kind 0 (the default) = numpy's legacy MT19937 stream, bit-exact with the reference.
extern "C" int mkb_sampler_set_rng(mkb_sampler_t *s, int kind, uint64_t seed, uint64_t draws) {
    MKB_REQUIRE(s != nullptr && (kind == 0 || kind == 1), "bad sampler / rng kind");
    MKB_REQUIRE(!s->drawn_ahead || kind == s->rng_kind, "a pool drawn ahead is pending: switch generators before the first generate");
    if (kind == 1) s->drawn_ahead = false;  // a pool drawn ahead from the old counter is discarded (as mkb_sampler_set_state does)
    s->rng_kind = kind; s->fast_seed = seed; s->fast_draw = draws;
    return MKB_OK;
}

extern "C" int mkb_sampler_get_rng(mkb_sampler_t *s, int *kind, uint64_t *seed, uint64_t *draws) {
    MKB_REQUIRE(s && kind && seed && draws, "null pointer");
    // (a pool drawn ahead has not been handed out yet: a resumed sampler must draw it again)
    *kind = s->rng_kind; *seed = s->fast_seed; *draws = s->fast_draw - ((s->rng_kind == 1 && s->drawn_ahead) ? 1 : 0);
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
    *D = s->draw_args(s->cur ^ 1, nullptr, /*save_prev=*/true);
    *lds_bytes = mkb::draw_lds_bytes(s->P(), s->P2());
    s->drawn_ahead = true;
    return true;
}

// First half of a generate: make buffer `cur` hold this batch's pool (drawn ahead, or drawn now by the stand-alone kernel).
static int take_pool(mkb_sampler *s, int64_t *pool_out, hipStream_t st, bool *was_ahead) {
    *was_ahead = s->drawn_ahead;
    s->drawn_ahead = false;
    s->cur ^= 1;
    if (!*was_ahead) {
        const mkb::DrawArgs D = s->draw_args(s->cur, pool_out, /*save_prev=*/false);
        hipLaunchKernelGGL(mkb::pool_draw_kernel, dim3(1), dim3(1024), mkb::draw_lds_bytes(D.P, D.P2), st, D);
        MKB_LAUNCH_CHECK();
    }
    return MKB_OK;
}

static mkb::FilterArgs filter_args(mkb_sampler *s, const int64_t *sample, int64_t B, int mode, int64_t *neg, int32_t *pos,
                                   uint16_t *cnt, int64_t *touched, int64_t *pool_out, int rw) {
    const bool head = mode == MKB_MODE_HEAD;
    const int P = s->P(), P2 = s->P2();
    mkb::FilterArgs F{};
    F.sample = sample; F.B = (int)B; F.head_mode = head ? 1 : 0;
    F.key_stride = head ? s->n_entity : s->n_relation;
    F.csr = head ? s->head : s->tail;
    F.pool = s->pool + (size_t)s->cur * P; F.lastflag = s->lastflag + (size_t)s->cur * P;
    F.sorted_val = s->sorted_val + (size_t)s->cur * P2; F.sorted_pos = s->sorted_pos + (size_t)s->cur * P2;
    F.K = (int)s->K; F.P = P; F.P2 = P2; F.rows_per_wg = rw;
    F.neg = neg; F.posmap = pos; F.cnt = cnt; F.touched = touched; F.pool_out = pool_out; F.status = s->status;
    return F;
}

int mkb::sampler_ride(mkb_sampler *s, const int64_t *sample, int64_t B, int mode, int64_t *neg, int64_t *pool, int32_t *pos,
                      uint16_t *cnt, int64_t *touched, FilterArgs *F, DrawArgs *D, const int64_t **pool_ids,
                      size_t *lds_bytes, hipStream_t st, int carrier_lanes) {
    MKB_REQUIRE(s && sample && neg, "null pointer");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "generate needs head-batch or tail-batch");
    MKB_REQUIRE(B > 0 && B <= INT32_MAX, "bad B");
    MKB_REQUIRE(s->P() <= 1024, "riding the optimizer launch supports size <= 512");
    bool was_ahead = false;
    if (int rc = take_pool(s, pool, st, &was_ahead)) return rc;
    // the carrier's blocks: one wave per row, as many rows per block as fit the carrier's 96 KB dynamic-LDS opt-in
    int rw = carrier_lanes / 64;
    if (const char *e = getenv("MKB_FILTER_RW")) rw = std::max(1, std::min(rw, atoi(e)));  // A/B switch (read per call)
    while (rw > 1 && filter_lds_bytes(s->P(), s->P2(), rw) > (size_t)96 * 1024) --rw;
    *F = filter_args(s, sample, B, mode, neg, pos, cnt, touched, was_ahead ? pool : nullptr, rw);
    *pool_ids = F->pool;
    *D = s->draw_args(s->cur ^ 1, nullptr, /*save_prev=*/true);  // the next pool, into the other buffer
    s->drawn_ahead = true;
    const size_t a = filter_lds_bytes(F->P, F->P2, rw), b = draw_lds_bytes(D->P, D->P2);
    *lds_bytes = a > b ? a : b;
    return MKB_OK;
}

extern "C" int mkb_sampler_generate(mkb_sampler_t *s, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                                    int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream) {
    MKB_REQUIRE(s && sample && neg, "null pointer");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "generate needs head-batch or tail-batch");
    MKB_REQUIRE(B >= 0 && B <= INT32_MAX, "bad B");
    hipStream_t st = (hipStream_t)stream;
    const int P = s->P(), P2 = s->P2();
    ProfScope ps(MKB_PROF_SAMPLER, st);
    bool ahead = false;  // the pool was drawn inside an earlier launch (sampler_draw_ahead / sampler_ride)
    if (int rc = take_pool(s, pool, st, &ahead)) return rc;
    if (B == 0) {
        if (ahead && pool)
            MKB_CHECK_HIP(hipMemcpyAsync(pool, s->pool + (size_t)s->cur * P, sizeof(int64_t) * (size_t)P, hipMemcpyDeviceToDevice, st));
        return MKB_OK;
    }
    const int rw = P <= 1024 ? 4 : 1;  // rows (waves) per workgroup: keeps the LDS request under 64 KB up to P = 2048
    const FilterArgs F = filter_args(s, sample, B, mode, neg, pos, cnt, touched, ahead ? pool : nullptr, rw);
    hipLaunchKernelGGL(filter_rows_kernel, dim3((unsigned)((B + rw - 1) / rw)), dim3(256), filter_lds_bytes(P, P2, rw), st, F);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

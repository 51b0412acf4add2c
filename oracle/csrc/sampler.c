/* Plain-C restatement of mkb.sampling.NegativeSampling.generate (TEST INFRASTRUCTURE ONLY).
 *
 * Follows /root/reference/mkb/sampling/negative_sampling.py:153-201 and the numpy 2.2.6
 * routines it calls (legacy RandomState MT19937 + masked-rejection randint; np.in1d with
 * assume_unique=True, invert=True: table / loop / sort paths).  Same semantics as
 * oracle/sampler.py, fast enough for full-size batches; used by tests and by bench.py's
 * cpu_baseline leg only.  Never linked into the product library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int32_t pos;
} orc_mt_t;

void orc_mt_seed(orc_mt_t *st, uint32_t seed) {
    for (int i = 0; i < MT_N; ++i) {
        st->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    st->pos = MT_N;
}

static void mt_regen(orc_mt_t *st) {
    uint32_t *mt = st->key;
    for (int i = 0; i < MT_N; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % MT_N] & 0x7fffffffu);
        mt[i] = mt[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    st->pos = 0;
}

uint32_t orc_mt_next(orc_mt_t *st) {
    if (st->pos == MT_N) mt_regen(st);
    uint32_t y = st->key[st->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void orc_randint(orc_mt_t *st, int64_t n, int64_t size, int64_t *out) {
    uint32_t rng = (uint32_t)(n - 1), mask = rng;
    if (rng == 0) { memset(out, 0, (size_t)size * sizeof(int64_t)); return; }
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t i = 0; i < size; ++i) {
        uint32_t v;
        do { v = orc_mt_next(st) & mask; } while (v > rng);
        out[i] = (int64_t)v;
    }
}

static int64_t lower_bound(const int64_t *a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* CSR filter: keys sorted ascending ([nk], key = a * n_entity_or_rel_stride + b as built by the
 * caller), offsets [nk+1], values sorted ascending inside each set.
 * Returns 0, -1 on unseen key (reference: KeyError), -2 if a row filters the whole pool
 * (reference: infinite loop). */
int orc_generate(orc_mt_t *st, int64_t n_entity, int64_t K, const int64_t *sample, int64_t B,
                 int head_mode, const int64_t *keys, int64_t nk, const int64_t *offsets,
                 const int64_t *values, int64_t key_stride, int64_t *neg_out, int64_t *pool_out) {
    const int64_t P = 2 * K;
    orc_randint(st, n_entity, P, pool_out);
    uint8_t *is_last = (uint8_t *)malloc((size_t)P);
    int64_t *f = (int64_t *)malloc((size_t)P * sizeof(int64_t));
    for (int64_t p = 0; p < P; ++p) {
        is_last[p] = 1;
        for (int64_t q = p + 1; q < P; ++q)
            if (pool_out[q] == pool_out[p]) { is_last[p] = 0; break; }
    }
    const double loop_thr = 10.0 * pow((double)P, 0.145);
    int rc = 0;
    for (int64_t i = 0; i < B && rc == 0; ++i) {
        int64_t h = sample[3 * i], r = sample[3 * i + 1], t = sample[3 * i + 2];
        int64_t key = head_mode ? r * key_stride + t : h * key_stride + r;
        int64_t ki = lower_bound(keys, nk, key);
        if (ki >= nk || keys[ki] != key) { rc = -1; break; }
        const int64_t *rec = values + offsets[ki];
        int64_t m = offsets[ki + 1] - offsets[ki];
        int64_t range = rec[m - 1] - rec[0];
        int sort_path = !(range <= 6 * (P + m)) && !((double)m < loop_thr);
        int64_t nf = 0;
        for (int64_t p = 0; p < P; ++p) {
            int64_t c = pool_out[p];
            int64_t j = lower_bound(rec, m, c);
            int member = (j < m && rec[j] == c);
            if (!member && (!sort_path || is_last[p])) f[nf++] = c;
        }
        if (nf == 0) { rc = -2; break; }
        for (int64_t j = 0; j < K; ++j) neg_out[i * K + j] = f[j % nf];
    }
    free(is_last);
    free(f);
    return rc;
}

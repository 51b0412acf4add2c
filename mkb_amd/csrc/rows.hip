// Row movement of the ROW-SHARDED entity table (mkb_amd/table_rows.py; BASELINE config 5).
//
// The reference is single-process: it has no counterpart to cite.  Rank g of `world` owns the entity rows e with
// e % world == g at shard index e / world.  A training step of one rank needs
//   * the batch's candidate pool rows (the same ids on every rank): every owner fills in the rows it holds, an
//     all-reduce of the (disjoint) block completes it;
//   * the heads and tails of its own triples, each from its owner: requests grouped by owner (mkb_rows_route), an
//     all-to-all of the ids, the owners read the rows (mkb_rows_gather), an all-to-all brings them back;
//   * afterwards the gradient rows go the same ways in reverse and the owners add them (mkb_rows_scatter_add).
// These kernels are the device side of that: plain HBM row copies, 16 bytes per lane, one workgroup per row, no host
// round trip (the split sizes of the all-to-alls are read back one batch AHEAD, see table_rows.py).
#include "common.h"

#include <stdlib.h>

namespace mkb {

constexpr int kRouteThreads = 1024;
constexpr int kRowThreads = 256;
constexpr int kMaxSegs = 4;

// ------------------------------------------------------------------------------------------------ route
struct RouteArgs {
    const int64_t *sample;  // [b, 3] (heads then tails are the 2b requests), or a flat id list [n] when flat != 0
    int b, world, flat, n_flat;
    int64_t row0;           // compact-table row of the first private row
    int64_t *send_ids;      // [2b] shard indices, grouped by owner, request order kept inside a group
    int32_t *slot;          // [2b] position of request j in the grouped order (j < b: head of triple j, else tail of j - b)
    int64_t *counts;        // [world]
    int64_t *compact;       // [b, 3] the triples re-addressed into the compact table: (row0 + slot[j], r, row0 + slot[b + j])
    int *bad;               // raised for a negative id (the reference raises IndexError, models/base.py:193-207); it is routed as id 0
    int merge_cap;          // slots of the LDS merge table (power of two >= 2 x requests), 0 = requests are not merged
};

// One workgroup.  Requests for the SAME row are merged first: a batch's heads and tails repeat (a hub entity is the head of
// dozens of a batch's triples; the synthetic YAGO3-10 of config 5 draws them Zipf-distributed), and every repeat used to be
// its own row of the compact table -- gathered, sent through the all-to-all, and its gradient row sent back and added with
// atomics at the owner.  Now request j takes the slot of the FIRST request for its row (an open-addressing table in LDS finds
// it: key = row id, value = smallest request index), the compact triples of the repeats point at that one row (the fused
// step's own occurrence counts then make its writers use atomics, exactly as on one GPU), and the lists that travel hold
// every row once.  Then a counting sort of the first requests by owner, done as `world` stable compactions (a few block
// scans of 1024 flags each; the list is 2b <= a few thousand ids, so this is latency, not bandwidth).  send_ids beyond the
// merged count are -1 (the row kernels skip such entries).
__global__ __launch_bounds__(kRouteThreads) void rows_route_kernel(RouteArgs A) {
    extern __shared__ __attribute__((aligned(8))) unsigned long long s_route[];  // merge table: [cap] keys, then [cap] ints
    __shared__ int s_cnt[kRouteThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = A.flat ? A.n_flat : 2 * A.b;
    const int cap = A.merge_cap;  // power of two >= 2 n, or 0: no merging (lists too long for the LDS table)
    unsigned long long *s_key = s_route;
    int *s_first = reinterpret_cast<int *>(s_route + cap);
    auto id_of = [&](int j) -> int64_t {
        int64_t id = A.flat ? A.sample[j] : (j < A.b ? A.sample[3 * (int64_t)j] : A.sample[3 * (int64_t)(j - A.b) + 2]);
        if (id < 0) {  // (no owner: the request still gets a slot -- row 0's --, the caller is told)
            if (A.bad) atomicOr(A.bad, 1);
            id = 0;
        }
        return id;
    };
    auto home = [&](int64_t id) { return (int)((((unsigned long long)id * 0x9E3779B97F4A7C15ull) >> 40) & (unsigned long long)(cap - 1)); };
    if (cap) {
        for (int e = tid; e < cap; e += kRouteThreads) { s_key[e] = ~0ull; s_first[e] = 0x7fffffff; }
        __syncthreads();
        for (int j = tid; j < n; j += kRouteThreads) {
            const unsigned long long id = (unsigned long long)id_of(j);
            for (int h = home((int64_t)id);; h = (h + 1) & (cap - 1)) {
                const unsigned long long prev = atomicCAS(&s_key[h], ~0ull, id);
                if (prev == ~0ull || prev == id) { atomicMin(&s_first[h], j); break; }
            }
        }
        __syncthreads();
    }
    auto first_of = [&](int j, int64_t id) {
        if (!cap) return j;
        for (int h = home(id);; h = (h + 1) & (cap - 1))
            if (s_key[h] == (unsigned long long)id) return s_first[h];
    };
    int offset = 0;
    for (int w = 0; w < A.world; ++w) {
        const int start = offset;
        for (int base = 0; base < n; base += kRouteThreads) {
            const int j = base + tid;
            int64_t id = 0;
            bool flag = false;
            if (j < n) {
                id = id_of(j);
                flag = (int)(id % A.world) == w && first_of(j, id) == j;
            }
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) s_cnt[wave] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int v = 0; v < kRouteThreads / 64; ++v) {
                const int c = s_cnt[v];
                before += v < wave ? c : 0;
                total += c;
            }
            if (flag) {
                const int sl = offset + before + __popcll(bal & ((1ull << lane) - 1ull));
                A.send_ids[sl] = id / A.world;
                A.slot[j] = sl;
            }
            offset += total;
            __syncthreads();
        }
        if (tid == 0) A.counts[w] = offset - start;
    }
    __threadfence_block();
    __syncthreads();  // the first requests' slots were written by other lanes of this workgroup
    for (int j = tid; j < n; j += kRouteThreads) {
        if (j >= offset) A.send_ids[j] = -1;  // (positions behind the merged list)
        if (cap) {
            const int f = first_of(j, id_of(j));
            if (f != j) A.slot[j] = A.slot[f];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (A.compact && !A.flat) {
        for (int j = tid; j < A.b; j += kRouteThreads) {
            A.compact[3 * (int64_t)j] = A.row0 + A.slot[j];
            A.compact[3 * (int64_t)j + 1] = A.sample[3 * (int64_t)j + 1];
            A.compact[3 * (int64_t)j + 2] = A.row0 + A.slot[A.b + j];
        }
    }
}

// ------------------------------------------------------------------------------------------------ gather / scatter
struct RowSeg {
    const int64_t *ids;
    float *rows;
    int64_t *local_ids;
    int n, world, rank;
};

struct RowMoveArgs {
    float *shard;        // gather: the table shard (read); scatter: its dense gradient (atomically added to)
    int64_t D, n_local;  // rows of the shard: an index outside [0, n_local) is skipped and reported through `bad`
    int *bad;
    RowSeg seg[kMaxSegs];
    int n_segs, row_blocks;
    // riders behind the row workgroups
    const float *weight; int n_weight; float *weight_sum;   // gather: sum of the batch's weights (fixed-order tree)
    float4 *zero; int64_t zero_vec4; int zero_tail;         // gather: clear a scratch buffer (the compact gradient)
    float *dense_dst; const float *dense_src; int64_t dense_n;  // scatter: dense_dst += dense_src (the relation gradient)
    float *copy_dst; const float *copy_src; int copy_n;         // scatter: copy_dst = copy_src (the loss leaves the ring buffers)
    int rider_blocks;
    // Exclusive rows.  occ[r] = how often shard row r is listed in this step's segments: counted by the gather launch (one
    // fire-and-forget atomic per listed row), read AND reset to 0 by the scatter launch of the same segments.  A row listed
    // once is added by ONE workgroup and by nobody else in that launch -> plain 16-byte read-modify-write instead of one L2
    // atomic per element (the atomic units retire ~1 element per clock and channel: 2.5 M of them were 45 us of the YAGO3-10
    // step).  A workgroup of a row listed several times may find the count already reset by a sibling: 0 != 1, atomics,
    // still correct; a stale count (a scatter that never ran) can only make a row look shared.
    unsigned *occ;
};

__device__ __forceinline__ bool locate(const RowMoveArgs &A, int block, int &s, int &j) {
    s = 0;
    j = block;
    while (s < A.n_segs && j >= A.seg[s].n) { j -= A.seg[s].n; ++s; }
    return s < A.n_segs;
}

// shard row of entry j of a segment, or -1 when another rank owns it (or when the index is outside the shard: an id the
// reference would answer with IndexError -- here the row is skipped and the caller's flag raised, never touched)
__device__ __forceinline__ int64_t shard_row(const RowMoveArgs &A, const RowSeg &S, int j) {
    const int64_t id = S.ids[j];
    int64_t r = id;
    if (S.world > 0) {
        if (id >= 0 && (int)(id % S.world) != S.rank) return -1;
        r = id < 0 ? id : id / S.world;
    } else if (id == -1) {
        return -1;  // (an entry the gather marked "not mine" in a local id list)
    }
    if (r < 0 || r >= A.n_local) {
        if (threadIdx.x == 0 && A.bad) atomicOr(A.bad, 2);
        return -1;
    }
    return r;
}

__global__ __launch_bounds__(kRowThreads) void rows_gather_kernel(RowMoveArgs A) {
    const int tid = threadIdx.x;
    int s, j;
    if ((int)blockIdx.x >= A.row_blocks) {
        const int rb = (int)blockIdx.x - A.row_blocks;
        if (rb == 0 && A.weight_sum) {  // fixed order: strided partials, wave butterfly, the four wave totals left to right
            __shared__ float red[kRowThreads / 64];
            float acc = 0.f;
            for (int i = tid; i < A.n_weight; i += kRowThreads) acc += A.weight[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();
            if (tid == 0) A.weight_sum[0] = red[0] + red[1] + red[2] + red[3];
        }
        for (int64_t e = (int64_t)rb * kRowThreads + tid; e < A.zero_vec4; e += (int64_t)A.rider_blocks * kRowThreads)
            A.zero[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rb == 0 && tid < A.zero_tail) reinterpret_cast<float *>(A.zero + A.zero_vec4)[tid] = 0.f;  // (odd row lengths)
        return;
    }
    if (!locate(A, (int)blockIdx.x, s, j)) return;
    const RowSeg &S = A.seg[s];
    const int64_t r = shard_row(A, S, j);
    if (tid == 0 && S.local_ids) S.local_ids[j] = r;
    if (tid == 0 && A.occ && r >= 0) atomicAdd(A.occ + r, 1u);
    float *out = S.rows + (int64_t)j * A.D;
    const float *in = A.shard + (r < 0 ? 0 : r) * A.D;
    if ((A.D & 3) == 0) {
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        float4 *out4 = reinterpret_cast<float4 *>(out);
        for (int64_t k = tid; k < (A.D >> 2); k += kRowThreads) out4[k] = r < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : in4[k];
    } else {
        for (int64_t k = tid; k < A.D; k += kRowThreads) out[k] = r < 0 ? 0.f : in[k];
    }
}

__global__ __launch_bounds__(kRowThreads) void rows_scatter_add_kernel(RowMoveArgs A) {
    const int tid = threadIdx.x;
    int s, j;
    if ((int)blockIdx.x >= A.row_blocks) {
        const int rb = (int)blockIdx.x - A.row_blocks;
        for (int64_t e = (int64_t)rb * kRowThreads + tid; e < A.dense_n; e += (int64_t)A.rider_blocks * kRowThreads)
            A.dense_dst[e] += A.dense_src[e];
        if (rb == 0 && tid < A.copy_n) A.copy_dst[tid] = A.copy_src[tid];
        return;
    }
    if (!locate(A, (int)blockIdx.x, s, j)) return;
    const RowSeg &S = A.seg[s];
    const int64_t r = shard_row(A, S, j);
    if (r < 0) return;
    const float *in = S.rows + (int64_t)j * A.D;
    float *g = A.shard + r * A.D;
    __shared__ unsigned s_cnt;
    if (A.occ) {
        if (tid == 0) {
            s_cnt = __hip_atomic_load(A.occ + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.occ[r] = 0u;  // ready for the next step's count
        }
        __syncthreads();
    }
    const bool own = A.occ && s_cnt == 1u;  // listed once in this step: nobody else writes the row in this launch
    if (own && (A.D & 3) == 0) {
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        float4 *g4 = reinterpret_cast<float4 *>(g);
        for (int64_t k = tid; k < (A.D >> 2); k += kRowThreads) {
            const float4 v = in4[k];
            float4 a = g4[k];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            g4[k] = a;
        }
        return;
    }
    if (own) {
        for (int64_t k = tid; k < A.D; k += kRowThreads) g[k] += in[k];
        return;
    }
    // duplicates (a pool id drawn twice, an entity that is several triples' head) add: one fp32 atomic per element
    if ((A.D & 3) == 0) {
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        for (int64_t k = tid; k < (A.D >> 2); k += kRowThreads) {
            const float4 v = in4[k];
            atomicAdd(g + 4 * k, v.x); atomicAdd(g + 4 * k + 1, v.y); atomicAdd(g + 4 * k + 2, v.z); atomicAdd(g + 4 * k + 3, v.w);
        }
    } else {
        for (int64_t k = tid; k < A.D; k += kRowThreads) atomicAdd(g + k, in[k]);
    }
}

static int fill_segs(RowMoveArgs &A, const mkb_row_seg_t *segs, int n_segs, int64_t D, bool need_rows16) {
    MKB_REQUIRE(n_segs >= 0 && n_segs <= kMaxSegs && (n_segs == 0 || segs), "0..%d row segments", kMaxSegs);
    int64_t total = 0;
    for (int s = 0; s < n_segs; ++s) {
        MKB_REQUIRE(segs[s].n >= 0 && segs[s].n <= INT32_MAX, "bad segment size");
        MKB_REQUIRE(segs[s].n == 0 || (segs[s].ids && segs[s].rows), "null segment pointer");
        MKB_REQUIRE(segs[s].world == 0 || (segs[s].world > 0 && segs[s].rank >= 0 && segs[s].rank < segs[s].world), "bad world / rank");
        MKB_REQUIRE(!need_rows16 || (D & 3) != 0 || (((uintptr_t)segs[s].rows) & 15) == 0, "row buffers must be 16-byte aligned");
        A.seg[s] = RowSeg{segs[s].ids, segs[s].rows, segs[s].local_ids, (int)segs[s].n, segs[s].world, segs[s].rank};
        total += segs[s].n;
    }
    MKB_REQUIRE(total <= INT32_MAX, "too many rows");
    A.n_segs = n_segs;
    A.row_blocks = (int)total;
    return MKB_OK;
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_rows_route(const int64_t *ids, int64_t n, int sample_layout, int world, int64_t row0, int64_t *send_ids,
                              int32_t *slot, int64_t *counts, int64_t *compact, int32_t *bad, void *stream) {
    MKB_REQUIRE(ids && send_ids && slot && counts, "null pointer");
    MKB_REQUIRE(n > 0 && 2 * n <= INT32_MAX && world >= 1 && world <= 4096, "bad n / world");
    MKB_REQUIRE(sample_layout || !compact, "the compact triples need the [b, 3] layout");
    RouteArgs A{ids, sample_layout ? (int)n : 0, world, sample_layout ? 0 : 1, (int)n, row0, send_ids, slot, counts, compact, bad, 0};
    const int64_t reqs = sample_layout ? 2 * n : n;
    int cap = 2;
    while (cap < 2 * reqs && cap <= 8192) cap <<= 1;
    static const bool no_merge = getenv("MKB_ROWS_NO_MERGE") != nullptr;  // A/B switch
    size_t lds = 0;
    if (cap <= 8192 && !no_merge) {  // 12 bytes per slot: 96 KB at the limit (4096 requests = 2048 triples per rank and step)
        A.merge_cap = cap;
        lds = (size_t)cap * 12;
        static LdsOptIn grant;
        if (grant.ensure(reinterpret_cast<const void *>(&rows_route_kernel), lds) != MKB_OK) {
            (void)hipGetLastError();  // a device that does not grant the table: route without merging (still correct, more rows travel)
            A.merge_cap = 0;
            lds = 0;
        }
    }
    hipLaunchKernelGGL(rows_route_kernel, dim3(1), dim3(kRouteThreads), lds, (hipStream_t)stream, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

extern "C" int mkb_rows_gather(const float *shard, int64_t n_local, int64_t D, const mkb_row_seg_t *segs, int n_segs,
                               const float *weight, int64_t n_weight, float *weight_sum, void *zero, int64_t zero_bytes,
                               uint32_t *occ, int32_t *bad, void *stream) {
    MKB_REQUIRE(shard && n_local > 0 && D > 0, "bad shard");
    MKB_REQUIRE((D & 3) != 0 || (((uintptr_t)shard) & 15) == 0, "shard must be 16-byte aligned");
    MKB_REQUIRE(!weight_sum || (weight && n_weight > 0 && n_weight <= INT32_MAX), "bad weights");
    MKB_REQUIRE(zero_bytes >= 0 && (zero_bytes & 3) == 0 && (zero_bytes == 0 || (zero && (((uintptr_t)zero) & 15) == 0)),
                "the buffer to clear must be 16-byte aligned and hold whole floats");
    RowMoveArgs A{};
    A.shard = const_cast<float *>(shard);
    A.D = D; A.n_local = n_local; A.bad = bad;
    if (int rc = fill_segs(A, segs, n_segs, D, true)) return rc;
    A.weight = weight; A.n_weight = (int)n_weight; A.weight_sum = weight_sum;
    A.occ = occ;
    A.zero = (float4 *)zero; A.zero_vec4 = zero_bytes >> 4; A.zero_tail = (int)((zero_bytes & 15) >> 2);
    int64_t rb = (A.zero_vec4 + 4 * kRowThreads - 1) / (4 * kRowThreads);  // ~4 float4 per lane
    if (rb > 1024) rb = 1024;
    if (rb < 1 && (weight_sum || A.zero_tail)) rb = 1;
    A.rider_blocks = (int)rb;
    if (A.row_blocks + A.rider_blocks == 0) return MKB_OK;
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)(A.row_blocks + A.rider_blocks)), dim3(kRowThreads), 0,
                       (hipStream_t)stream, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

extern "C" int mkb_rows_scatter_add(float *grad, int64_t n_local, int64_t D, const mkb_row_seg_t *segs, int n_segs,
                                    float *dense_dst, const float *dense_src, int64_t dense_n, float *copy_dst,
                                    const float *copy_src, int64_t copy_n, uint32_t *occ, int32_t *bad, void *stream) {
    MKB_REQUIRE(grad && n_local > 0 && D > 0, "bad gradient shard");
    MKB_REQUIRE(dense_n >= 0 && (dense_n == 0 || (dense_dst && dense_src)), "bad dense rider");
    MKB_REQUIRE(copy_n >= 0 && copy_n <= kRowThreads && (copy_n == 0 || (copy_dst && copy_src)), "bad copy rider");
    RowMoveArgs A{};
    A.shard = grad;
    A.D = D; A.n_local = n_local; A.bad = bad;
    if (int rc = fill_segs(A, segs, n_segs, D, true)) return rc;
    A.dense_dst = dense_dst; A.dense_src = dense_src; A.dense_n = dense_n;
    A.copy_dst = copy_dst; A.copy_src = copy_src; A.copy_n = (int)copy_n;
    A.occ = occ;
    int64_t rb = (dense_n + 4 * kRowThreads - 1) / (4 * kRowThreads);
    if (rb > 256) rb = 256;
    if (rb == 0 && copy_n > 0) rb = 1;
    A.rider_blocks = (int)rb;
    if (A.row_blocks + A.rider_blocks == 0) return MKB_OK;
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3((unsigned)(A.row_blocks + A.rider_blocks)), dim3(kRowThreads), 0,
                       (hipStream_t)stream, A);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

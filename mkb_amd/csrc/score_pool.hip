// Pooled scoring path: host driver, row kernels and the C-ABI entry points (mkb_pool_step / mkb_pool_score_fwd /
// mkb_pool_score_bwd).  The three tile kernels live in score_pool_kernels.h and are instantiated per model in
// score_pool_<model>.hip (parallel compilation); see the header for the design.
#include "score_pool_kernels.h"
#include "gemm_mfma.h"

#include <stdlib.h>

namespace mkb {

// ------------------------------------------------------------------------------------------------ row kernels
struct RowArgs {
    const float *ent, *rel;
    const int64_t *sample;
    float *Q;          // [B, De] out (build) / [nslices, B, De] dQ partials in (backward)
    float *g_ent, *g_rel;
    int64_t De, Dr;
    int d, B, nslices;
    float kd;
    // GEMM route: depth[i] = one past the last pool position row i uses (cnt [B, P]); null = not wanted
    const uint16_t *cnt;
    int *depth;
    int P;
};

// depth[i] for the GEMM route's "used pool depth" cuts (gemm_mfma.h): one workgroup per row scans the row's multiplicities
__device__ __forceinline__ void row_depth_256(const uint16_t *__restrict__ cnt, int P, int64_t i, int *__restrict__ depth) {
    __shared__ int s_dep[4];
    int m = 0;
    for (int p = threadIdx.x; p < P; p += 256)
        if (cnt[i * P + p]) m = p + 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_dep[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) depth[i] = max(max(s_dep[0], s_dep[1]), max(s_dep[2], s_dep[3]));
}

// Q[i] = query of row i (same math as the LDS staging of the general forward kernel, written to global)
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void query_build_kernel(RowArgs A) {
    const int64_t i = blockIdx.x;
    if (A.depth) row_depth_256(A.cnt, A.P, i, A.depth);
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];
    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    float *q = A.Q + i * A.De;
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        const float *e = HEAD ? et : eh;
        for (int u = threadIdx.x; u < A.d; u += 256) {
            Cplx qq = build_q_cplx<MODEL, HEAD>(Cplx{e[u], e[A.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f}, A.kd);
            q[u] = qq.re;
            q[A.d + u] = qq.im;
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256)
            q[u] = build_q_real<MODEL, HEAD>(HEAD ? er[u] : eh[u], HEAD ? et[u] : er[u], A.kd);
    }
}

// sum of the slice partials of dQ for N elements of one row: the loads of four slices x N elements are requested together
// and added in slice order.  (A plain loop over a run-time slice count waits for every load before it issues the next: the
// ISA of row_bwd had 8 serial L2 round trips per lane and loop iteration there -- 32 per workgroup, most of the launch.)
template <int N>
__device__ __forceinline__ void dq_slices_sum(const float *__restrict__ dq, int64_t sstride, int nslices, const int (&k)[N], float (&out)[N]) {
#pragma unroll
    for (int e = 0; e < N; ++e) out[e] = 0.f;
    for (int s0 = 0; s0 < nslices; s0 += 4) {
        float v[4][N];
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const float *src = dq + (int64_t)min(s0 + z, nslices - 1) * sstride;
#pragma unroll
            for (int e = 0; e < N; ++e) v[z][e] = src[k[e]];
        }
#pragma unroll
        for (int z = 0; z < 4; ++z)
#pragma unroll
            for (int e = 0; e < N; ++e) out[e] += s0 + z < nslices ? v[z][e] : 0.f;
    }
}

// chain dQ[i] (sum of the slice partials) into the fixed operands' gradient rows (duplicate rows add: atomics)
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void query_bwd_kernel(RowArgs A) {
    const int64_t i = blockIdx.x;
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];
    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    const float *dq = A.Q + i * A.De;
    const int64_t sstride = (int64_t)A.B * A.De;
    float *g_e = A.g_ent + (HEAD ? t : h) * A.De;
    float *g_r = A.g_rel + r * A.Dr;
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        const float *e = HEAD ? et : eh;
        for (int u = threadIdx.x; u < A.d; u += 256) {
            Cplx de, dr;
            float dqn[2];
            dq_slices_sum<2>(dq, sstride, A.nslices, {u, A.d + u}, dqn);
            query_bwd_cplx<MODEL, HEAD>(Cplx{e[u], e[A.d + u]}, Cplx{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f},
                                        Cplx{dqn[0], dqn[1]}, A.kd, de, dr);
            atomicAdd(g_e + u, de.re);
            atomicAdd(g_e + A.d + u, de.im);
            atomicAdd(g_r + u, dr.re);
            if constexpr (MODEL == MKB_COMPLEX) atomicAdd(g_r + A.d + u, dr.im);
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256) {
            float da, db, dqn[1];
            dq_slices_sum<1>(dq, sstride, A.nslices, {u}, dqn);
            query_bwd_real<MODEL, HEAD>(HEAD ? er[u] : eh[u], HEAD ? et[u] : er[u], dqn[0], A.kd, da, db);
            atomicAdd((HEAD ? g_r : g_e) + u, da);
            atomicAdd((HEAD ? g_e : g_r) + u, db);
        }
    }
}

// ---- fused row kernels of the training step --------------------------------------------------------------
// One workgroup per batch row, lanes own units.  They replace four launches of the step (general forward and
// backward for the positive triple, query build, query backward) by two, read h / r / t rows once, and merge
// the positive- and negative-path contributions to the same table row before the (single) atomic per element.
struct RowStepArgs {
    const float *ent, *rel, *modulus;
    const int64_t *sample;
    float *Q;                 // [B, De] negative-path queries (out of row_fwd)
    const float *dQ;          // [nslices, B, De] partials (in of row_bwd)
    float *pos_score;         // [B] out of row_fwd
    const float *dpos;        // [B] d loss / d pos_score (in of row_bwd)
    float *g_ent, *g_rel, *g_modulus;
    int64_t De, Dr;
    int d, B, nslices;
    float kd, gamma;
    // row_bwd inside mkb_pool_step: its last workgroup also finishes the loss (adversarial_finish_block), or nullptr
    const float *loss_rowpart, *loss_scal;
    float *loss_out;
    // ... and extra workgroups behind the B row workgroups reduce the dx partials of the single-pass backward (blocks > 0)
    DxReduce dx;
    // ... or (MFMA path) add the split-K partials of the scattered product into the table gradient (kind 2: sc.M workgroups)
    GemmTail sc;
    // few relations (WN18RR: 11, YAGO3-10: 37): ~B / R rows of a batch add into the SAME gradient row, and same-line atomics
    // serialise in L2.  Row i then adds into copy (i % rel_copies) of a [copies, R, Dr] scratch that the row forward zeroed;
    // rel_fold_kernel adds the copies into g_rel behind the row backward.  rel_copies <= 1: straight into g_rel.
    float *rel_rep;
    int rel_copies, n_rel;
    // Exclusive rows.  occ[e] = how often entity e occurs among this batch's pool ids, heads and tails: zeroed for exactly
    // those entries by row_fwd, counted by the pooled forward (MFMA path: by the loss kernel), read by row_bwd.  An h / t
    // row that occurs once is written by
    // ONE workgroup of the row backward and by nobody else in that launch (the riders only write pool rows): it takes a
    // plain read-modify-write instead of one L2 atomic per element (the L2 atomic units retire ~1 element per clock and
    // channel: row_bwd was bound by them; 72 % of the heads and 57 % of the tails of an FB15k-237 batch occur once).
    int *occ;
    const int64_t *pool;
    int P;
    // GEMM route: depth[i] = one past the last pool position row i uses, written by row_fwd (see RowArgs::depth)
    const uint16_t *cnt;
    int *depth;
    int depth_P;
    // (Round 4 tried 16 bytes per lane with every load of a row issued before the first use -- 125 VGPRs, half the resident
    // waves, atomics at stride 16: row_bwd 29 -> 53 us at the headline shape.  The row kernels are bound by bytes and by
    // the L2 atomic units, not by the latency of their loads: 32 waves per CU hide that.  Removed.)
    // row_bwd: the gradient rows this step writes are known to be all-zero (the row-lazy optimizer's advance launch consumed
    // and cleared them): rows written by exactly one workgroup are STORED, not read-modified-written
    int grads_clear;
};

__device__ __forceinline__ float block_sum_256_row(float v, float *red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// positive score (mode None == tail-style formula against the true tail, pipeline.py:211) + Q for the negatives
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void row_fwd_kernel(RowStepArgs A) {
    __shared__ float red[4];
    const int64_t i = blockIdx.x;
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];

    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    float *q = A.Q + i * A.De;
    float acc = 0.f;
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        for (int u = threadIdx.x; u < A.d; u += 256) {
            const Cplx ch{eh[u], eh[A.d + u]}, ct{et[u], et[A.d + u]};
            const Cplx cr{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f};
            const Cplx qp = build_q_cplx<MODEL, false>(ch, cr, A.kd);
            const Cplx qn = HEAD ? build_q_cplx<MODEL, true>(ct, cr, A.kd) : qp;
            q[u] = qn.re;
            q[A.d + u] = qn.im;
            if constexpr (ModelTraits<MODEL>::cplx_pair) acc += pair_term_cmod(qp, ct);
            else acc += pair_term_real<MODEL, false>(qp.re, ct.re, A.kd) + pair_term_real<MODEL, false>(qp.im, ct.im, A.kd);
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256) {
            const float vh = eh[u], vr = er[u], vt = et[u];
            const float qp = build_q_real<MODEL, false>(vh, vr, A.kd);
            q[u] = HEAD ? build_q_real<MODEL, true>(vr, vt, A.kd) : qp;
            acc += pair_term_real<MODEL, false>(qp, vt, A.kd);
        }
    }
    acc = block_sum_256_row(acc, red);
    if (threadIdx.x == 0) A.pos_score[i] = finish_score<MODEL>(acc, A.gamma, MODEL == MKB_PROTATE ? A.modulus[0] : 0.f);
    if (A.depth) row_depth_256(A.cnt, A.depth_P, i, A.depth);
    // housekeeping for the later kernels of the step, behind the row's own work (fire-and-forget stores)
    if (A.occ && threadIdx.x == 64) {  // (counted by the pooled forward / the loss kernel, read by the row backward)
        A.occ[h] = 0; A.occ[t] = 0;
        for (int64_t p = i; p < A.P; p += A.B) A.occ[A.pool[p]] = 0;
    }
    if (A.rel_copies > 1)  // the row backward's relation-gradient copies start from zero
        for (int64_t e = i * 256 + threadIdx.x; e < (int64_t)A.rel_copies * A.n_rel * A.Dr; e += (int64_t)A.B * 256) A.rel_rep[e] = 0.f;
}

// backward of the positive pair + chain of both query gradients into the rows of h, r, t
template <int MODEL, bool HEAD>
__global__ __launch_bounds__(256) void row_bwd_kernel(RowStepArgs A) {
    __shared__ float red[4];
    // rider: dx partial reduction of the pooled backward (one launch fewer per step).  Its workgroups come FIRST: they are
    // few and heavy (8 partial rows each); dispatched behind the 1024 row workgroups they formed a 30 us tail.
    if ((int)blockIdx.x < A.dx.blocks) {
        pool_dx_reduce_block(A.dx, (int)blockIdx.x);
        return;
    }
    const int sc_blocks = A.sc.kind == 2 ? A.sc.M : 0;
    if ((int)blockIdx.x < A.dx.blocks + sc_blocks) {
        __shared__ int s_red[16];
        splitk_scatter_block(A.sc.part, A.sc.out, A.sc.c_idx, A.sc.M, A.sc.N, A.sc.ldc, A.sc.nz, (int)blockIdx.x - A.dx.blocks,
                             A.sc.depth, A.sc.n_depth, s_red);
        return;
    }
    const int64_t i = (int64_t)blockIdx.x - A.dx.blocks - sc_blocks;
    const int64_t h = A.sample[3 * i], r = A.sample[3 * i + 1], t = A.sample[3 * i + 2];
    const float *eh = A.ent + h * A.De, *er = A.rel + r * A.Dr, *et = A.ent + t * A.De;
    float *g_h = A.g_ent + h * A.De, *g_t = A.g_ent + t * A.De;
    float *g_r = A.rel_copies > 1 ? A.rel_rep + ((int64_t)(i % A.rel_copies) * A.n_rel + r) * A.Dr : A.g_rel + r * A.Dr;
    const float *dq = A.dQ + i * A.De;
    const int64_t sstride = (int64_t)A.B * A.De;
    const float gp = A.dpos[i];
    const bool own_h = A.occ && A.occ[h] == 1, own_t = A.occ && A.occ[t] == 1;  // workgroup-uniform
    // rows one workgroup alone writes: a plain store when the row is known to be all-zero (grads_clear), else read-modify-
    // write; shared rows: one fp32 atomic per element
    const bool st_h = own_h && A.grads_clear, st_t = own_t && A.grads_clear;  // workgroup-uniform
    auto add_h = [&](int k, float v) { if (st_h) g_h[k] = v; else if (own_h) g_h[k] += v; else atomicAdd(g_h + k, v); };
    auto add_t = [&](int k, float v) { if (st_t) g_t[k] = v; else if (own_t) g_t[k] += v; else atomicAdd(g_t + k, v); };
    const float modulus = (MODEL == MKB_PROTATE) ? A.modulus[0] : 0.f;
    float extra = 0.f;
    if constexpr (ModelTraits<MODEL>::cplx_query) {
        for (int u = threadIdx.x; u < A.d; u += 256) {
            const Cplx ch{eh[u], eh[A.d + u]}, ct{et[u], et[A.d + u]};
            const Cplx cr{er[u], MODEL == MKB_COMPLEX ? er[A.d + u] : 0.f};
            // positive pair: q = h (x) rot, candidate = t
            const Cplx qp = build_q_cplx<MODEL, false>(ch, cr, A.kd);
            Cplx dqp, dxp;
            if constexpr (ModelTraits<MODEL>::cplx_pair) {
                pair_bwd_cmod(qp, ct, gp, dqp, dxp);
            } else {
                float e0 = 0.f;
                pair_bwd_real<MODEL, false>(qp.re, ct.re, gp, A.kd, modulus, dqp.re, dxp.re, e0);
                pair_bwd_real<MODEL, false>(qp.im, ct.im, gp, A.kd, modulus, dqp.im, dxp.im, e0);
            }
            Cplx dh, dr, dt = dxp, de, dr2;
            query_bwd_cplx<MODEL, false>(ch, cr, dqp, A.kd, dh, dr);
            // negative path: q = conj(rot) (x) t (head-batch) or h (x) rot (tail-batch)
            float dqs[2];
            dq_slices_sum<2>(dq, sstride, A.nslices, {u, A.d + u}, dqs);
            const Cplx dqn{dqs[0], dqs[1]};
            query_bwd_cplx<MODEL, HEAD>(HEAD ? ct : ch, cr, dqn, A.kd, de, dr2);
            if constexpr (HEAD) { dt.re += de.re; dt.im += de.im; } else { dh.re += de.re; dh.im += de.im; }
            dr.re += dr2.re; dr.im += dr2.im;
            add_h(u, dh.re); add_h(A.d + u, dh.im);
            add_t(u, dt.re); add_t(A.d + u, dt.im);
            atomicAdd(g_r + u, dr.re);
            if constexpr (MODEL == MKB_COMPLEX) atomicAdd(g_r + A.d + u, dr.im);
        }
    } else {
        for (int u = threadIdx.x; u < (int)A.De; u += 256) {
            const float vh = eh[u], vr = er[u], vt = et[u];
            const float qp = build_q_real<MODEL, false>(vh, vr, A.kd);
            float dqp, dxp, e0 = 0.f;
            pair_bwd_real<MODEL, false>(qp, vt, gp, A.kd, modulus, dqp, dxp, e0);
            extra += gp * e0;
            float dh, dr, dt = dxp;
            query_bwd_real<MODEL, false>(vh, vr, dqp, A.kd, dh, dr);  // tail-style: a = h, b = r
            float da, db, dqs[1];
            dq_slices_sum<1>(dq, sstride, A.nslices, {u}, dqs);
            query_bwd_real<MODEL, HEAD>(HEAD ? vr : vh, HEAD ? vt : vr, dqs[0], A.kd, da, db);
            if constexpr (HEAD) { dr += da; dt += db; } else { dh += da; dr += db; }
            add_h(u, dh);
            atomicAdd(g_r + u, dr);
            add_t(u, dt);
        }
    }
    if constexpr (MODEL == MKB_PROTATE) {  // d score / d modulus = - sum_k |sin z| for the positive pair
        extra = block_sum_256_row(extra, red);
        if (threadIdx.x == 0) atomicAdd(A.g_modulus, -extra);
    }
    if (A.loss_out && i == A.B - 1)  // the per-row loss terms were written by an earlier kernel
        adversarial_finish_block(A.loss_rowpart, A.B, A.loss_scal, A.loss_out, red);
}

// g_rel[e] += sum_c rep[c][e]   (fixed order; rep was zeroed by this step's loss kernel)
__global__ __launch_bounds__(256) void rel_fold_kernel(const float *__restrict__ rep, float *__restrict__ g_rel, int copies, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    // eight copies' loads at a time (a loop with a run-time trip count waits for every load before it issues the next:
    // 23 copies were 23 round trips -- 7 us for half a megabyte); added up in copy order
    float s = 0.f;
    const float old = g_rel[e];
    for (int c0 = 0; c0 < copies; c0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = rep[(int64_t)min(c0 + k, copies - 1) * n + e];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += c0 + k < copies ? v[k] : 0.f;
    }
    g_rel[e] = old + s;
}

__global__ __launch_bounds__(256) void masked_copy_kernel(const float *__restrict__ src, const uint16_t *__restrict__ cnt,
                                                          float *__restrict__ dst, int64_t n, int64_t n_pad, int64_t P, SeedLayout SL) {
    // n = B * P entries of the batch; [n, n_pad): the rows that pad the last tile of 8 in the blocked layout, written as 0
    // (the dense pass of the single-pass backward reads a tile's 8 seeds without looking at B)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[seed_index(SL, i / P, i % P, P)] = cnt[i] ? src[i] : 0.f;
    else if (i < n_pad) dst[seed_index(SL, i / P, i % P, P)] = 0.f;
}

// ------------------------------------------------------------------------------------------------ host side
struct Workspace {
    float *Q, *dQ, *G, *dpos, *scratch, *gemm_part, *dXp;
    unsigned long long *xused;
    float *rel_rep;
    int *occ;
    int *depth;  // [B] used pool depth per row (GEMM route)
    float *tile_part;
    size_t bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int units_of(const mkb_tables_t *tb) { return tb->model == MKB_ROTATE ? tb->hidden_dim : (int)tb->entity_dim; }

// Kernel configuration for a table shape: units per lane (vector width of the loads), waves per workgroup, and how
// many workgroups share a row tile / position tile so that the grid fills 256 CUs with 16-32 waves each.
// ComplEx / DistMult: the pair function is a dot product, so the pooled block is three dense fp32 GEMMs on the matrix
// cores (gemm_mfma.h) instead of the lane-owns-dims VALU kernels.  MKB_POOL_NO_MFMA=1 keeps the VALU kernels (A/B).
// (The two per-call switches -- MKB_POOL_NO_MFMA, MKB_POOL_DENSE -- change the workspace layout.  Callers cache a workspace
// per table shape, so mkb_pool_step_workspace_bytes reports the LARGEST layout over the switches' settings: g_force_* let it
// ask pick_config for each.  Found in round 4 by filling freed device memory with NaN between calls: a workspace sized for the
// matrix route was handed to the VALU route of the same shape -- tests that flipped the switch had been reading and writing past it.)
static thread_local int g_force_mfma = -1, g_force_dense = -1;  // -1: as the environment says

static bool use_mfma(const mkb_tables_t *tb) {
    const char *e = getenv("MKB_POOL_NO_MFMA");  // read per call: the tests switch it within one process
    const bool off = g_force_mfma >= 0 ? g_force_mfma == 0 : (e && e[0] == '1');
    return !off && (tb->model == MKB_COMPLEX || tb->model == MKB_DISTMULT) && tb->entity_dim >= 16;
}

static bool pick_config(const mkb_tables_t *tb, int64_t B, int64_t P, PoolLaunch &L) {
    // relation-gradient copies of the row backward: when a relation averages >= 16 rows of the batch, spread them so that
    // ~4 rows share a copy, within 1 MB of scratch (WN18RR B = 1024: 11 relations -> 23 copies; FB15k-237: none)
    L.rel_elems = tb->n_relation * tb->relation_dim;
    L.rel_copies = 1;
    L.n_entity = tb->n_entity;
    {
        static const bool off = getenv("MKB_POOL_NO_REL_COPIES") != nullptr;  // A/B switch
        const int64_t per_rel = tb->n_relation > 0 ? B / tb->n_relation : 0;
        if (!off && per_rel >= 16 && L.rel_elems > 0) {
            int64_t c = per_rel / 4;
            const int64_t cap = (1 << 18) / L.rel_elems;  // 1 MB of floats
            if (c > cap) c = cap;
            if (c > 64) c = 64;
            if (c >= 2) L.rel_copies = (int)c;
        }
    }
    const int NU = units_of(tb);
    const int64_t De = tb->entity_dim, d = tb->hidden_dim;
    const bool cp = tb->model == MKB_ROTATE;
    const bool al16 = (((uintptr_t)tb->ent) & 15) == 0;
    const bool even2 = al16 && NU % 2 == 0 && De % 2 == 0 && (!cp || d % 2 == 0);
    const bool even4 = al16 && NU % 4 == 0 && De % 4 == 0 && (!cp || d % 4 == 0);
    // units per lane: vector loads + packed math need even dims; waves: smallest workgroup that covers the row
    const bool k4 = even4 && tb->model != MKB_PROTATE;  // (pRotatE: no 4-units-per-lane instantiations, see launch_head)
    if (even2 && NU >= 64 && NU <= 2048) L.kpt = 2;
    else if (k4 && NU > 2048 && NU <= 4096) L.kpt = 4;
    else if (NU <= 1024) L.kpt = 1;
    else return false;
    const int lanes = (NU + L.kpt - 1) / L.kpt;
    if (L.kpt == 1) L.nw = lanes <= 64 ? 1 : (lanes <= 128 ? 2 : (lanes <= 256 ? 4 : 16));
    else if (L.kpt == 2) L.nw = lanes <= 64 ? 1 : (lanes <= 128 ? 2 : (lanes <= 256 ? 4 : (lanes <= 512 ? 8 : 16)));
    else L.nw = 16;
    if (const char *e = getenv("MKB_POOL_CFG")) {  // experiment knob: "kpt,nw"
        int k = 0, n = 0;
        if (sscanf(e, "%d,%d", &k, &n) == 2 && (k == 1 || (k == 2 && even2) || (k == 4 && even4)) && (int64_t)k * n * 64 >= NU) {
            L.kpt = k; L.nw = n;
        }
    }
    // forward: 4 units per lane when the row allows it (measured 91 -> 77 us at the headline shape: the per-position
    // wave reduction is amortised over twice the pair evaluations); the backward kernels gain nothing from it
    L.fkpt = L.kpt; L.fnw = L.nw;
    if (k4 && NU > 512 && NU <= 1024) { L.fkpt = 4; L.fnw = 4; }
    // Forward of the complex-modulus models: dense prefix [0, Kd) of the pool on the outer-product register tile, the sparse
    // fringe on the row-tile kernel in the same launch (score_pool_tile.h).  Needs the 4-wave forward configuration (rows of
    // 257 .. 1024 complex dims), whole 64-position tiles in the prefix (K = P / 2 >= 64) and 16-byte rows.
    L.tile = 0; L.tile_kd = 0; L.tile_ks = 1; L.tile_fringe_slices = 1;
    {
        static const bool off = getenv("MKB_POOL_TILE") && getenv("MKB_POOL_TILE")[0] == '0';  // A/B switch
        const int64_t Kd = (P / 2) / 64 * 64;
        // (round 5: TransE's |q - x| on the same tile -- its "dims" are pairs of floats, so that a chunk is 32 floats either way)
        const bool te = tb->model == MKB_TRANSE && De % 4 == 0;
        const int64_t dt = cp ? d : De / 2;
        if (!off && (te || (cp && d % 4 == 0)) && !use_mfma(tb) && al16 && L.fnw == 4 && (L.fkpt == 4 || L.fkpt == 2) && Kd >= 64 && B >= 64) {
            L.tile = 1;
            L.tile_kd = (int)Kd;
            const int tiles = (int)((B + 63) / 64) * (int)(Kd / 64);
            int ks = 1;
            while (tiles * ks < 512 && ks < 16 && dt / (ks * 2) >= 32) ks *= 2;  // fill the chip; >= 2 chunks of 16 dims per split
            if (const char *e = getenv("MKB_POOL_TILE_KS")) { const int v = atoi(e); if (v >= 1 && v <= 16) ks = v; }
            L.tile_ks = ks;
            L.tile_fringe_slices = 2;
            if (const char *e = getenv("MKB_POOL_TILE_FSL")) { const int v = atoi(e); if (v >= 1 && v <= 16) L.tile_fringe_slices = v; }
        }
    }
    const int target = 256 * 16 / L.nw;  // workgroups for ~16 waves per CU
    const int row_tiles = (int)((B + TI - 1) / TI), pos_tiles = (int)((P + TI - 1) / TI);
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    // Slice counts, swept on the headline shape and on the dimension shards of 2 / 4 / 8 GPUs (tools/shard_emulate.py:
    // rows grow, rows get shorter, workgroups shrink to 4 / 2 / 1 waves).
    // forward: ~32 waves per CU (short rows: 99 -> 79 us at the 8-GPU shard; flat at the headline shape)
    L.fwd_slices = clampi((256 * 32 / L.fnw + row_tiles - 1) / row_tiles, 1, kMaxSlices);
    L.mfma = use_mfma(tb) ? 1 : 0;
    // dq pass: it shares its launch with the dx pass, so with big workgroups half the waves suffice (headline: 2 slices
    // instead of 4 keep the merged kernel at 160 us and save row_bwd 4 us of partial-buffer reads); 1- and 2-wave
    // workgroups want the full count (8-GPU shard: 210 -> 185 us).  GEMM route: complete dQ rows.
    const int q_target = L.nw >= 4 ? target / 2 : target;
    L.q_slices = L.mfma ? 1 : clampi((q_target + row_tiles - 1) / row_tiles, 1, kMaxSlices);
    // x pass: rows of a slice are listed in LDS (40 B each): keep a slice <= 256 rows so several workgroups fit a CU;
    // single-wave workgroups do better with 256-row slices than with 128 (8-GPU shard: 209 -> 192 us)
    const int min_x = (int)((B + 255) / 256);
    const int x_target = L.nw == 1 ? target / 2 : target;
    L.x_slices = clampi((x_target + pos_tiles - 1) / pos_tiles, min_x > 2 ? min_x : 2, 1 << 20);
    // Single-pass backward (pool_bwd1_kernel; VALU models only): dims are cut into slices of 64 * kpt units, positions into
    // blocks of <= 64 * halves (LDS accumulator <= 128 KB), row tiles into groups of 16 waves.  Position blocks are doubled
    // until the grid holds ~4096 waves (16 per CU); each block costs one dQ partial buffer.
    {
        const char *e = getenv("MKB_POOL_BWD1");  // A/B switch: 0 = the two-pass merged kernel
        L.bwd1 = (!L.mfma && !(e && e[0] == '0')) ? 1 : 0;
        L.row_groups = 0; L.cplx = cp ? 1 : 0; L.pb_halves = 0; L.tiles_per_wave = 1;
        // units per lane of the single-pass backward: 2 (1 for odd rows); real-valued models with long rows take 4 -- the
        // same 4 floats per lane and position as RotatE's two complex dims, half the per-position bookkeeping of 2
        static const bool no_k4 = getenv("MKB_POOL_BWD1_NO_K4") != nullptr;  // A/B switch
        const int k1 = (!cp && k4 && NU >= 512 && !no_k4) ? 4 : (L.kpt >= 2 ? 2 : 1), nc = k1 * (cp ? 2 : 1);
        L.bkpt = k1;
        const int lanes1 = (NU + k1 - 1) / k1;
        L.dim_slices = (lanes1 + 63) / 64;
        const int max_halves = 128 * 1024 / (64 * nc * 64 * 4);  // 2 at nc = 4, 4 at nc = 2, 8 at nc = 1
        int npb = 1;
        while (((P + npb - 1) / npb + 63) / 64 > max_halves) npb *= 2;
        while (npb < kMaxSlices && (int64_t)row_tiles * L.dim_slices * npb < 4096 && (P + npb - 1) / npb > 32) npb *= 2;
        if (const char *q = getenv("MKB_POOL_PBLOCKS")) { const int v = atoi(q); if (v >= npb && v <= kMaxSlices && (v & (v - 1)) == 0) npb = v; }
        if (npb > kMaxSlices) L.bwd1 = 0;
        if (L.bwd1) {
            L.q_slices = npb;
            // halves: ROUNDED UP to a power of two.  The kernel cuts its 16 chunks into 16 / halves lanes per half, the seed
            // layout (common.h) shifts by log2(halves), and carve() sizes G / dXp with the same value: 3, 5, 6, 7 would break
            // all three (e.g. TransE hidden 500, B 2048, K 384: 192 positions per block = 3 halves).  max_halves is a power
            // of two, so the rounded value still fits the LDS accumulator; slots past P are masked in the kernel.
            int halves = (int)(((P + npb - 1) / npb + 63) / 64), h2 = 1;
            while (h2 < halves) h2 *= 2;
            L.pb_halves = h2;
            if (h2 > max_halves || (npb & (npb - 1)) != 0) L.bwd1 = 0;  // (cannot happen: both loops above keep the invariants)
        }
        L.dense_lanes = 0;
        if (L.bwd1) {
            // Dense pass (pool_bwd1_kernel<..., DENSE>).  Every row takes its first K = P / 2 surviving candidates, so positions
            // p < K are used by (nearly) every row.  Slot (h, l) holds position block + npb * (l * halves + h): lanes
            // l < K / (npb * halves) of every half are dense; rounded down to whole lanes per chunk (a chunk = every cph-th lane
            // of a half).  Round 3: the same gradients; per step, same box, general vs dense pass:
            // headline 0.242 -> 0.240 ms, WN18RR 0.150 -> 0.143, YAGO3-10 0.206 -> 0.202, TransE-1000 0.183 -> 0.185.  So: on
            // for the complex-modulus pair function; the real-valued ones have no dense form (round 4: their ten instantiations
            // were 0.6 MB of the library for a path that measured slower); MKB_POOL_DENSE=0 switches it off (read per call: the
            // tests switch it within one process).
            const char *e = getenv("MKB_POOL_DENSE");
            const int cph = 16 / L.pb_halves;
            const int ld = (int)((P / 2) / ((int64_t)npb * L.pb_halves)) / cph * cph;
            // (round 5: TransE as well -- with its pair term down to 5 VALU operations the general pass's per-position bookkeeping is a
            // quarter of the loop: 62.7 -> 59.5 us at the headline shape, same call; DistMult / ComplEx / pRotatE never had the form)
            const bool dense_model = cp || tb->model == MKB_TRANSE;
            const bool on = dense_model && (g_force_dense >= 0 ? g_force_dense == 1 : (e ? e[0] == '1' : true));  // (launch_bwd1 compiles the dense form for the complex-modulus models only)
            if (on && cph >= 1 && ld > 0 && ld <= 32) L.dense_lanes = ld;
        }
        // Small problems: when the single-pass grid would be a handful of 16-wave workgroups, its ring and its prologue ARE the
        // launch (Umls TransE-64, K 16, B 256: two workgroups, 43 us for 0.5 MFLOP).  One wave per (row tile, <= 64 positions,
        // 64 * kpt units) instead: pool_bwd_wave_kernel; position slices until ~1024 waves are in flight.  MKB_POOL_SMALL=0: A/B.
        L.small = 0; L.skpt = 1; L.schunks = 1;
        if (L.bwd1) {
            const char *e = getenv("MKB_POOL_SMALL");
            const int groups1 = (row_tiles + 15) / 16;
            const int skpt = L.kpt >= 2 ? 2 : 1;
            const int chunks = (NU + 64 * skpt - 1) / (64 * skpt);
            int nsl = (int)((P + 63) / 64);
            while (nsl < kMaxSlices && (int64_t)row_tiles * chunks * nsl < 1024 && P / nsl > 2) ++nsl;
            const int limit = e && atoi(e) > 1 ? atoi(e) : 32;  // (MKB_POOL_SMALL=<n>: the workgroup count below which it applies)
            if (!(e && e[0] == '0') && (int64_t)groups1 * npb * L.dim_slices <= limit && nsl <= kMaxSlices) {
                L.small = 1; L.bwd1 = 0; L.skpt = skpt; L.schunks = chunks; L.q_slices = nsl;
                L.dense_lanes = 0; L.pb_halves = 0;
            }
        }
        if (L.bwd1) {
            const int64_t waves = (int64_t)row_tiles * L.dim_slices * npb;
            L.tiles_per_wave = (int)(waves >= 3 * 4096 ? waves / (2 * 4096) : 1);
            if (const char *q = getenv("MKB_POOL_TPW")) { const int v = atoi(q); if (v >= 1 && v <= 64) L.tiles_per_wave = v; }
            L.row_groups = (row_tiles + 16 * L.tiles_per_wave - 1) / (16 * L.tiles_per_wave);
            L.cplx = cp ? 1 : 0;
        }
    }
    if (const char *e = getenv("MKB_POOL_FSLICES")) { const int v = atoi(e); if (v >= 1 && v <= 64) L.fwd_slices = v; }
    if (const char *e = getenv("MKB_POOL_QSLICES")) { const int v = atoi(e); if (v >= 1 && v <= kMaxSlices && !L.mfma && !L.bwd1 && !L.small) L.q_slices = v; }
    if (const char *e = getenv("MKB_POOL_XSLICES")) { const int v = atoi(e); if (v >= min_x && v >= 1) L.x_slices = v; }
    return true;
}

static Workspace carve(void *ws, int64_t B, int64_t P, int64_t De, const PoolLaunch &L) {
    Workspace w;
    unsigned char *p = (unsigned char *)ws;
    size_t off = 0;
    auto take = [&](size_t n) { void *r = p ? p + off : nullptr; off += align256(n); return (float *)r; };
    w.Q = take((size_t)B * De * 4);
    w.dQ = take((size_t)(L.mfma ? kMfmaDqSlices : L.q_slices) * B * De * 4);  // (MFMA route: room for the dQ product's K split)
    // gradient seeds: plain [B, P], or the tile-blocked layout of the single-pass backward (rows padded to tiles of 8,
    // positions to blocks * halves * 64 slots)
    const size_t g_plain = (size_t)B * P, g_blocked = L.bwd1 ? (size_t)((B + 7) / 8) * L.q_slices * L.pb_halves * 64 * 8 : 0;
    w.G = take((g_plain > g_blocked ? g_plain : g_blocked) * 4);
    w.dpos = take((size_t)B * 4);
    w.scratch = take((size_t)(B + 1) * 4);
    w.gemm_part = take(L.mfma ? (size_t)8 * B * (P > De ? P : De) * 4 : 0);  // split-K partials of the MFMA path
    // single-pass backward: dx partials per row group [groups][blocks][slots][dim slices][64 lanes][NC] + used-slot masks
    const size_t nc = (size_t)L.bkpt * (L.cplx ? 2 : 1);
    w.dXp = take(L.bwd1 ? (size_t)L.row_groups * L.q_slices * L.pb_halves * 64 * L.dim_slices * 64 * nc * 4 : 0);
    w.xused = (unsigned long long *)take(L.bwd1 ? (size_t)L.row_groups * L.q_slices * 8 * 8 : 0);
    w.rel_rep = take(L.rel_copies > 1 ? (size_t)L.rel_copies * L.rel_elems * 4 : 0);
    w.occ = (int *)take((size_t)L.n_entity * 4);
    w.depth = (int *)take(L.mfma ? (size_t)B * 4 : 0);
    w.tile_part = take(L.tile ? (size_t)L.tile_ks * B * L.tile_kd * 4 : 0);  // dim-split partial scores of the dense prefix
    w.bytes = off;
    return w;
}

#ifdef MKB_TRACE_WG
static unsigned long long *g_trace = nullptr;
static int g_trace_kind = 2;
extern "C" void mkb_debug_set_trace(void *p, int kind) { g_trace = (unsigned long long *)p; g_trace_kind = kind; }
#endif

static SeedLayout seed_layout(const PoolLaunch &L) {
    if (!L.bwd1) return SeedLayout{-1, 0};
    auto lg = [](int v) { int b = 0; while ((1 << b) < v) ++b; return b; };
    return SeedLayout{lg(L.q_slices), lg(L.pb_halves)};
}

static PoolArgs make_args(const mkb_tables_t *tb, const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t P,
                          const Workspace &w, const PoolLaunch &L) {
    PoolArgs A{};
    A.ent = tb->ent; A.Q = w.Q; A.pool = pool; A.cnt = cnt; A.G = w.G; A.dQ = w.dQ;
    A.B = (int)B; A.P = (int)P; A.d = tb->hidden_dim; A.De = tb->entity_dim; A.kd = tb->phase_div;
    A.modulus = tb->modulus; A.x_slices = L.x_slices; A.q_slices = L.q_slices;
    A.dXp = w.dXp; A.xused = w.xused;
    A.g_blocked = L.bwd1 ? 1 : 0;
    const bool g = tb->model == MKB_TRANSE || tb->model == MKB_ROTATE || tb->model == MKB_PROTATE;
    A.c0 = g ? tb->gamma : 0.f;
    A.c1 = g ? -1.f : 1.f;
#ifdef MKB_TRACE_WG
    A.trace = g_trace; A.trace_kind = g_trace_kind;
#endif
    return A;
}

static pool_launch_fn launcher_of(int model) {
    switch (model) {
        case MKB_TRANSE: return pool_launch_transe;
        case MKB_ROTATE: return pool_launch_rotate;
        case MKB_COMPLEX: return pool_launch_complex;
        case MKB_DISTMULT: return pool_launch_distmult;
        case MKB_PROTATE: return pool_launch_protate;
    }
    return nullptr;
}

template <int MODEL, bool HEAD>
static int run_query_build(const RowArgs &ra, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL((query_build_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int run_query_bwd(const RowArgs &ra, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL((query_bwd_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int run_row_fwd(const RowStepArgs &ra, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL((row_fwd_kernel<MODEL, HEAD>), dim3((unsigned)B), dim3(256), 0, st, ra);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

template <int MODEL, bool HEAD>
static int run_row_bwd(const RowStepArgs &ra, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL((row_bwd_kernel<MODEL, HEAD>), dim3((unsigned)(B + ra.dx.blocks + (ra.sc.kind == 2 ? ra.sc.M : 0))), dim3(256),
                       0, st, ra);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

#define MKB_DISPATCH(fn, model, head, ...)                                                            \
    switch (model) {                                                                                  \
        case MKB_TRANSE: return (head) ? fn<MKB_TRANSE, true>(__VA_ARGS__) : fn<MKB_TRANSE, false>(__VA_ARGS__);       \
        case MKB_ROTATE: return (head) ? fn<MKB_ROTATE, true>(__VA_ARGS__) : fn<MKB_ROTATE, false>(__VA_ARGS__);       \
        case MKB_COMPLEX: return (head) ? fn<MKB_COMPLEX, true>(__VA_ARGS__) : fn<MKB_COMPLEX, false>(__VA_ARGS__);    \
        case MKB_DISTMULT: return (head) ? fn<MKB_DISTMULT, true>(__VA_ARGS__) : fn<MKB_DISTMULT, false>(__VA_ARGS__); \
        case MKB_PROTATE: return (head) ? fn<MKB_PROTATE, true>(__VA_ARGS__) : fn<MKB_PROTATE, false>(__VA_ARGS__);    \
    }                                                                                                 \
    return set_error(MKB_ERR_INVALID, "unknown model")

static int dispatch_query_build(const mkb_tables_t *tb, bool head, const RowArgs &ra, int64_t B, hipStream_t st) {
    MKB_DISPATCH(run_query_build, tb->model, head, ra, B, st);
}
static int dispatch_query_bwd(const mkb_tables_t *tb, bool head, const RowArgs &ra, int64_t B, hipStream_t st) {
    MKB_DISPATCH(run_query_bwd, tb->model, head, ra, B, st);
}

static int dispatch_row_fwd(const mkb_tables_t *tb, bool head, const RowStepArgs &ra, int64_t B, hipStream_t st) {
    MKB_DISPATCH(run_row_fwd, tb->model, head, ra, B, st);
}
static int dispatch_row_bwd(const mkb_tables_t *tb, bool head, const RowStepArgs &ra, int64_t B, hipStream_t st) {
    MKB_DISPATCH(run_row_bwd, tb->model, head, ra, B, st);
}

static int pooled_fwd(const mkb_tables_t *tb, bool head, const int64_t *sample, const int64_t *pool, const uint16_t *cnt,
                      int64_t B, int64_t P, float *S, const Workspace &w, const PoolLaunch &L, hipStream_t st,
                      bool build_queries = true, GemmTail *s_tail = nullptr, int *occ = nullptr, bool *occ_counted = nullptr) {
    if (s_tail) s_tail->kind = 0;
    if (occ_counted) *occ_counted = false;
    static const bool no_cut = getenv("MKB_GEMM_NO_DEPTH") != nullptr;  // A/B: multiply all P pool positions
    const bool cut = use_mfma(tb) && !no_cut;
    if (build_queries) {
        RowArgs ra{tb->ent, tb->rel, sample, w.Q, nullptr, nullptr, tb->entity_dim, tb->relation_dim, tb->hidden_dim, (int)B,
                   1, tb->phase_div};
        if (cut) { ra.cnt = cnt; ra.depth = w.depth; ra.P = (int)P; }
        if (int rc = dispatch_query_build(tb, head, ra, B, st)) return rc;
    }
    if (use_mfma(tb)) {  // S = Q . ent[pool]^T on the matrix cores, over the pool positions the row tile uses (cnt masks later)
        GemmArgs g{};
        g.A = w.Q; g.lda = tb->entity_dim; g.B = tb->ent; g.ldb = tb->entity_dim; g.b_idx = pool; g.b_rows = tb->n_entity;
        g.C = S; g.ldc = P; g.M = (int)B; g.N = (int)P; g.K = (int)tb->entity_dim; g.c0 = 0.f; g.c1 = 1.f;
        if (cut) { g.depth = w.depth; g.depth_mode = 1; g.n_depth = (int)B; }
        ProfScope ps(MKB_PROF_POOL_FWD, st);
        return launch_gemm<true, true, GEMM_STORE_AFFINE>(g, st, w.gemm_part, s_tail, s_tail ? 500 : 0);
    }
    PoolArgs A = make_args(tb, pool, cnt, B, P, w, L);  // (the kernel also zero-fills the entries no row uses)
    A.S = S;
    if (occ) { A.occ = occ; A.occ_sample = sample; if (occ_counted) *occ_counted = true; }
    ProfScope ps(MKB_PROF_POOL_FWD, st);
    if (L.tile) {  // (s_tail: the loss rows of mkb_pool_step add the prefix's partial sums up; otherwise S is finished here)
        A.tile_part = w.tile_part;
        A.tile_tail = s_tail;
        return launcher_of(tb->model)(5, head, L, A, st);
    }
    return launcher_of(tb->model)(0, head, L, A, st);
}

// dq_slices (out): how many [B, De] partial products the dQ buffer holds (the consumer -- row / query backward -- adds them up)
static int pooled_bwd(const mkb_tables_t *tb, bool head, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                      const uint16_t *cnt, int64_t B, int64_t P, const Workspace &w, const PoolLaunch &L, hipStream_t st,
                      bool chain_queries = true, DxReduce *dx_out = nullptr, GemmTail *x_tail = nullptr, int *dq_slices = nullptr) {
    int dq_used = L.q_slices;
    if (x_tail) x_tail->kind = 0;
    static const bool no_cut = getenv("MKB_GEMM_NO_DEPTH") != nullptr;
    const bool cut = use_mfma(tb) && !no_cut;  // w.depth was written by this call's row_fwd / query_build
    if (use_mfma(tb)) {
        // dQ [B, De] = G [B, P] . ent[pool]
        GemmArgs gq{};
        gq.A = w.G; gq.lda = P; gq.B = tb->ent; gq.ldb = tb->entity_dim; gq.b_idx = pool; gq.b_rows = tb->n_entity;
        gq.C = w.dQ; gq.ldc = tb->entity_dim; gq.M = (int)B; gq.N = (int)tb->entity_dim; gq.K = (int)P;
        if (cut) { gq.depth = w.depth; gq.depth_mode = 2; gq.n_depth = (int)B; }  // (G is exactly 0 beyond a row's depth)
        // g_ent[pool[p]] += (G^T [P, B] . Q [B, De])[p]
        GemmArgs gx{};
        gx.A = w.G; gx.lda = P; gx.B = w.Q; gx.ldb = tb->entity_dim; gx.b_idx = nullptr;
        gx.C = gr->g_ent; gx.ldc = tb->entity_dim; gx.c_idx = pool; gx.M = (int)P; gx.N = (int)tb->entity_dim; gx.K = (int)B;
        if (cut) { gx.depth = w.depth; gx.depth_mode = 3; gx.n_depth = (int)B; }
        bool paired = false;
        {   // round 5: both products in ONE launch where they qualify (profiled as the POOL_BWD_Q class)
            ProfScope ps(MKB_PROF_POOL_BWD_Q, st);
            if (int rc = launch_gemm_bwd_pair(gq, kMfmaDqSlices, &dq_used, gx, w.gemm_part, x_tail, st, &paired)) return rc;
        }
        if (!paired) {
            {
                ProfScope ps(MKB_PROF_POOL_BWD_Q, st);
                // a K split of this product leaves its partials in the dQ slices: the row backward sums them (it does so for the
                // VALU route's position blocks anyway) instead of a reduction launch in between (DistMult: 6.5 us)
                if (int rc = launch_gemm<true, false, GEMM_STORE>(gq, st, w.gemm_part, nullptr, 0, kMfmaDqSlices, &dq_used)) return rc;
            }
            ProfScope ps(MKB_PROF_POOL_BWD_X, st);
            if (int rc = launch_gemm<false, false, GEMM_ATOMIC_ROWS>(gx, st, w.gemm_part, x_tail, x_tail ? 500 : 0)) return rc;
        }
    } else {
        PoolArgs A = make_args(tb, pool, cnt, B, P, w, L);
        A.g_modulus = gr->g_modulus;
        A.g_ent = gr->g_ent;
        A.dx_reduce_out = dx_out;
        static const bool split = getenv("MKB_POOL_SPLIT_BWD") != nullptr;  // A/B: the two passes as two launches
        if (L.bwd1) {  // every pair term evaluated once (pool_bwd1_kernel); profiled as the POOL_BWD_Q class
            ProfScope ps(MKB_PROF_POOL_BWD_Q, st);
            if (int rc = launcher_of(tb->model)(4, head, L, A, st)) return rc;
        } else if (L.small) {  // a small problem: one wave per piece, every pair evaluated once (pool_bwd_wave_kernel)
            ProfScope ps(MKB_PROF_POOL_BWD_Q, st);
            if (int rc = launcher_of(tb->model)(6, head, L, A, st)) return rc;
        } else if (!split) {  // dq and dx passes in one grid (pool_bwd_kernel); profiled as the POOL_BWD_Q class
            ProfScope ps(MKB_PROF_POOL_BWD_Q, st);
            if (int rc = launcher_of(tb->model)(1, head, L, A, st)) return rc;
        } else {
            {
                ProfScope ps(MKB_PROF_POOL_BWD_Q, st);
                if (int rc = launcher_of(tb->model)(3, head, L, A, st)) return rc;
            }
            ProfScope ps(MKB_PROF_POOL_BWD_X, st);
            if (int rc = launcher_of(tb->model)(2, head, L, A, st)) return rc;
        }
    }
    if (dq_slices) *dq_slices = dq_used;
    if (chain_queries) {
        RowArgs ra{tb->ent, tb->rel, sample, w.dQ, gr->g_ent, gr->g_rel, tb->entity_dim, tb->relation_dim, tb->hidden_dim,
                   (int)B, dq_used, tb->phase_div};
        if (int rc = dispatch_query_bwd(tb, head, ra, B, st)) return rc;
    }
    return MKB_OK;
}

static int check_pool_call(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                           int64_t K, int mode, const void *ws, PoolLaunch &L) {
    if (int rc = validate_tables(tb)) return rc;
    MKB_REQUIRE(sample && pool && cnt && ws, "null pointer");
    MKB_REQUIRE(B > 0 && K > 0 && B <= INT32_MAX, "bad B / K");
    MKB_REQUIRE(tb->n_entity <= INT32_MAX, "n_entity too large");
    MKB_REQUIRE(mode == MKB_MODE_HEAD || mode == MKB_MODE_TAIL, "the pooled path needs head-batch or tail-batch");
    MKB_REQUIRE((((uintptr_t)ws) & 255) == 0, "workspace must be 256-byte aligned");
    if (2 * K > kMaxP || !pick_config(tb, B, 2 * K, L))
        return set_error(MKB_ERR_UNSUPPORTED, "shape not covered by the pooled kernels (size <= 1024, rows <= 4096 units, "
                                              "even dims above 256 units); use the general path");
    return MKB_OK;
}

}  // namespace mkb

using namespace mkb;

extern "C" int mkb_pool_supported(const mkb_tables_t *tb, int64_t B, int64_t K) {
    PoolLaunch L;
    return tb && B > 0 && K > 0 && 2 * K <= kMaxP && tb->n_entity <= INT32_MAX && pick_config(tb, B, 2 * K, L);
}

extern "C" int64_t mkb_pool_step_workspace_bytes(const mkb_tables_t *tb, int64_t B, int64_t K) {
    if (!tb || B <= 0 || K <= 0) return 0;
    int64_t best = 0;
    bool any = false;
    for (int fm = 0; fm <= 1; ++fm)
        for (int fd = 0; fd <= 1; ++fd) {  // the largest layout over the per-call switches (see g_force_mfma)
            PoolLaunch L;
            g_force_mfma = fm; g_force_dense = fd;
            const bool ok = pick_config(tb, B, 2 * K, L);
            const int64_t n = ok ? (int64_t)carve(nullptr, B, 2 * K, tb->entity_dim, L).bytes : 0;
            g_force_mfma = -1; g_force_dense = -1;
            any = any || ok;
            if (n > best) best = n;
        }
    PoolLaunch L;
    if (!pick_config(tb, B, 2 * K, L)) return 0;  // (the shape must be supported as the environment stands)
    return any ? best : 0;
}

extern "C" int mkb_pool_score_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt,
                                  int64_t B, int64_t K, int mode, float *pool_score, void *ws, void *stream) {
    PoolLaunch L;
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws, L)) return rc;
    MKB_REQUIRE(pool_score != nullptr, "pool_score is null");
    const Workspace w = carve(ws, B, 2 * K, tb->entity_dim, L);
    return pooled_fwd(tb, mode_is_head(mode), sample, pool, cnt, B, 2 * K, pool_score, w, L, (hipStream_t)stream);
}

extern "C" int mkb_pool_score_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                                  const uint16_t *cnt, int64_t B, int64_t K, int mode, const float *dpool_score, void *ws,
                                  void *stream) {
    PoolLaunch L;
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws, L)) return rc;
    MKB_REQUIRE(gr && gr->g_ent && gr->g_rel && dpool_score, "null pointer");
    MKB_REQUIRE(tb->model != MKB_PROTATE || gr->g_modulus, "pRotatE needs g_modulus");
    Workspace w = carve(ws, B, 2 * K, tb->entity_dim, L);
    hipStream_t st = (hipStream_t)stream;
    // rebuild the queries (the forward's copy may have been overwritten by another call sharing the workspace)
    RowArgs ra{tb->ent, tb->rel, sample, w.Q, nullptr, nullptr, tb->entity_dim, tb->relation_dim, tb->hidden_dim, (int)B, 1,
               tb->phase_div};
    if (use_mfma(tb)) { ra.cnt = cnt; ra.depth = w.depth; ra.P = (int)(2 * K); }  // used pool depth per row (GEMM cuts)
    if (int rc = dispatch_query_build(tb, mode_is_head(mode), ra, B, st)) return rc;
    // G = caller's gradient with the entries no row uses forced to 0 (the single-pass backward reads the mask off G)
    const SeedLayout sl = seed_layout(L);
    const int64_t n_pad = sl.log2_blocks >= 0 ? (B + 7) / 8 * 8 * 2 * K : B * 2 * K;
    hipLaunchKernelGGL(masked_copy_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, dpool_score, cnt, w.G,
                       B * 2 * K, n_pad, 2 * K, sl);
    MKB_LAUNCH_CHECK();
    return pooled_bwd(tb, mode_is_head(mode), gr, sample, pool, cnt, B, 2 * K, w, L, st);
}

// The two halves of mkb_pool_step.  Between them the caller may combine the scores of several devices that each
// hold a slice of the embedding DIMENSIONS (mkb_amd.parallel.DimSharded*): scores are sums over dims, so the halves'
// only coupling is pos_score / pool_score.
// s_tail (mkb_pool_step only): the MFMA forward may leave the scores as split-K partials for the loss rows to reduce
static int pool_step_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                         int64_t K, int mode, float *pos_score, float *pool_score, void *ws, void *stream, GemmTail *s_tail) {
    PoolLaunch L;
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws, L)) return rc;
    MKB_REQUIRE(pos_score && pool_score, "null pointer");
    const int64_t P = 2 * K;
    const Workspace w = carve(ws, B, P, tb->entity_dim, L);
    hipStream_t st = (hipStream_t)stream;
    const bool head = mode_is_head(mode);
    RowStepArgs ra{tb->ent, tb->rel, tb->modulus, sample, w.Q, w.dQ, pos_score, w.dpos, nullptr, nullptr, nullptr,
                   tb->entity_dim, tb->relation_dim, tb->hidden_dim, (int)B, L.q_slices, tb->phase_div, tb->gamma};
    static const bool no_own = getenv("MKB_POOL_NO_OWN") != nullptr;  // A/B: every gradient row through atomics
    if (!no_own) { ra.occ = w.occ; ra.pool = pool; ra.P = (int)P; }
    if (L.rel_copies > 1) { ra.rel_rep = w.rel_rep; ra.rel_copies = L.rel_copies; ra.n_rel = (int)tb->n_relation; }
    static const bool no_cut = getenv("MKB_GEMM_NO_DEPTH") != nullptr;
    if (use_mfma(tb) && !no_cut) { ra.cnt = cnt; ra.depth = w.depth; ra.depth_P = (int)P; }  // used pool depth per row (GEMM cuts)
    // positive pass (mode None: tail-style formula against the true tail, pipeline.py:211) + negative-path queries
    {
        ProfScope ps(MKB_PROF_GENERAL_FWD, st);
        if (int rc = dispatch_row_fwd(tb, head, ra, B, st)) return rc;
    }
    // negative pass over the shared pool (pipeline.py:230-232)
    return pooled_fwd(tb, head, sample, pool, cnt, B, P, pool_score, w, L, st, /*build_queries=*/false,
                      (s_tail && P <= 64 * 16) ? s_tail : nullptr, ra.occ);
}

// s_tail: scores still in split-K partials (from pool_step_fwd); fold: the scattered product's tail rides the row backward
static int pool_step_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                         const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                         const float *weight_sum, const float *pos_score, float *pool_score, float *loss, void *ws,
                         void *stream, const GemmTail *s_tail) {
    PoolLaunch L;
    if (int rc = check_pool_call(tb, sample, pool, cnt, B, K, mode, ws, L)) return rc;
    MKB_REQUIRE(gr && gr->g_ent && gr->g_rel && weight && pos_score && pool_score && loss, "null pointer");
    MKB_REQUIRE(tb->model != MKB_PROTATE || gr->g_modulus, "pRotatE needs g_modulus");
    const int64_t P = 2 * K;
    const Workspace w = carve(ws, B, P, tb->entity_dim, L);
    hipStream_t st = (hipStream_t)stream;
    const bool head = mode_is_head(mode);
    RowStepArgs ra{tb->ent, tb->rel, tb->modulus, sample, w.Q, w.dQ, nullptr, w.dpos, gr->g_ent, gr->g_rel, gr->g_modulus,
                   tb->entity_dim, tb->relation_dim, tb->hidden_dim, (int)B, L.q_slices, tb->phase_div, tb->gamma};
    // Adversarial forward + gradient seeds (pipeline.py:234 and the head of :236)
    static const bool no_own = getenv("MKB_POOL_NO_OWN") != nullptr;
    if (!no_own) { ra.occ = w.occ; ra.pool = pool; ra.P = (int)P; }
    if (L.rel_copies > 1) {  // (zeroed by the row forward kernel of this step)
        ra.rel_rep = w.rel_rep; ra.rel_copies = L.rel_copies; ra.n_rel = (int)tb->n_relation;
    }
    // (the occurrence counts: the VALU forward kernel counted them; the MFMA path has no such kernel, the loss rows do it)
    if (int rc = adversarial_launch(pos_score, pool_score, weight, cnt, B, P, alpha, weight_sum, loss, w.dpos, w.G, w.scratch, st,
                                    /*defer_finish=*/true, seed_layout(L), s_tail, use_mfma(tb) ? ra.occ : nullptr,
                                    sample, pool)) return rc;
    ra.loss_rowpart = w.scratch + 1;
    ra.loss_scal = weight_sum ? weight_sum : w.scratch;
    ra.loss_out = loss;
    // backward (pipeline.py:236): pooled negatives, then the positive pair and both query chains in one row kernel
    static const bool no_fold = getenv("MKB_GEMM_NO_FOLD") != nullptr;
    if (int rc = pooled_bwd(tb, head, gr, sample, pool, cnt, B, P, w, L, st, /*chain_queries=*/false, L.bwd1 ? &ra.dx : nullptr,
                            no_fold ? nullptr : &ra.sc, &ra.nslices)) return rc;
    ra.dx.occ = ra.occ;  // (the dx reduction riding this launch writes pool rows: exclusive ones without atomics)
    ra.grads_clear = gr->rows_clear ? 1 : 0;
    ra.dx.clear = ra.grads_clear;
    ProfScope ps(MKB_PROF_GENERAL_BWD, st);
    if (int rc = dispatch_row_bwd(tb, head, ra, B, st)) return rc;
    if (ra.rel_rep) {
        hipLaunchKernelGGL(rel_fold_kernel, dim3((unsigned)((L.rel_elems + 255) / 256)), dim3(256), 0, st, ra.rel_rep, gr->g_rel,
                           L.rel_copies, L.rel_elems);
        MKB_LAUNCH_CHECK();
    }
    return MKB_OK;
}

extern "C" int mkb_pool_step_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt,
                                 int64_t B, int64_t K, int mode, float *pos_score, float *pool_score, void *ws,
                                 void *stream) {
    return pool_step_fwd(tb, sample, pool, cnt, B, K, mode, pos_score, pool_score, ws, stream, nullptr);
}

extern "C" int mkb_pool_step_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                                 const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                                 const float *weight_sum, const float *pos_score, const float *pool_score, float *loss,
                                 void *ws, void *stream) {
    return pool_step_bwd(tb, gr, sample, weight, pool, cnt, B, K, mode, alpha, weight_sum, pos_score,
                         const_cast<float *>(pool_score), loss, ws, stream, nullptr);
}

extern "C" int mkb_pool_step(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                             const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                             const float *weight_sum, float *pos_score, float *pool_score, float *loss, void *ws,
                             void *stream) {
    // one call: the split-K tails of the MFMA products fold into the launches that follow them (MKB_GEMM_NO_FOLD=1: A/B)
    static const bool no_fold = getenv("MKB_GEMM_NO_FOLD") != nullptr;
    GemmTail s_tail{};
    if (int rc = pool_step_fwd(tb, sample, pool, cnt, B, K, mode, pos_score, pool_score, ws, stream, no_fold ? nullptr : &s_tail))
        return rc;
    return pool_step_bwd(tb, gr, sample, weight, pool, cnt, B, K, mode, alpha, weight_sum, pos_score, pool_score, loss, ws,
                         stream, (s_tail.kind == 1 || s_tail.kind == 3) ? &s_tail : nullptr);
}

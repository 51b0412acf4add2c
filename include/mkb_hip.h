/* mkb_hip.h -- C ABI of libmkb_hip.so: the MI355X (gfx950) kernels behind mkb's triplet-scoring,
 * self-adversarial loss and filtered negative-sampling hot path.
 *
 * The reference (raphaelsty/mkb) is pure Python on PyTorch and has NO FFI / plugin interface
 * (SURVEY.md 8b); the boundary it offers is its public Python API.  Each entry point below names the
 * reference call it stands in for (paths relative to /root/reference).  The host side
 * (the mkb_amd Python package) mirrors that Python API and calls these symbols through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller (Python glue) owns and
 *     allocates every buffer, nothing is allocated or freed across the ABI except opaque handles;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); every call is asynchronous with
 *     respect to the host and ordered on that stream;
 *   - return value: 0 = ok, <0 = mkb_status_t error; mkb_last_error() gives the message (thread local);
 *   - floating point is IEEE fp32, indices are int64 (torch.LongTensor) unless stated;
 *   - tables are row-major contiguous: ent [n_entity, entity_dim], rel [n_relation, relation_dim];
 *     RotatE / ComplEx rows hold the real half first, then the imaginary half (models/rotate.py:76-77).
 */
#ifndef MKB_HIP_H
#define MKB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKB_ABI_VERSION 6

typedef enum {
    MKB_OK = 0,
    MKB_ERR_INVALID = -1,   /* bad argument (shape, enum, null pointer, misalignment)            */
    MKB_ERR_HIP = -2,       /* a HIP runtime call failed                                          */
    MKB_ERR_KEY = -3,       /* sampler: (r,t) / (h,r) not in the training triples (ref: KeyError) */
    MKB_ERR_EMPTY = -4,     /* sampler: a row's filter removed the whole pool (ref: infinite loop)*/
    MKB_ERR_UNSUPPORTED = -5
} mkb_status_t;

/* models/transe.py:65, rotate.py:69, complex.py:65, distmult.py:63, protate.py:74 */
typedef enum { MKB_TRANSE = 0, MKB_ROTATE = 1, MKB_COMPLEX = 2, MKB_DISTMULT = 3, MKB_PROTATE = 4 } mkb_model_t;

/* models/base.py:153-164: any mode string other than the two below selects the default batch */
typedef enum { MKB_MODE_DEFAULT = 0, MKB_MODE_HEAD = 1, MKB_MODE_TAIL = 2 } mkb_mode_t;

/* Parameter set of one model == the nn.Parameters of models/base.py:66-100 (+ modulus, protate.py:72). */
typedef struct {
    int32_t model;            /* mkb_model_t */
    int32_t hidden_dim;
    int64_t n_entity, n_relation;
    int64_t entity_dim, relation_dim;
    const float *ent;         /* [n_entity, entity_dim]                       */
    const float *rel;         /* [n_relation, relation_dim]                   */
    const float *modulus;     /* [1] device scalar (pRotatE), may be null     */
    float gamma;              /* gamma.item()                                 */
    float phase_div;          /* fp32(embedding_range.item() / pi): rotate.py:79, protate.py:78-80 */
} mkb_tables_t;

/* Dense gradient buffers == .grad of the same parameters (accumulated into, never zeroed here). */
typedef struct {
    float *g_ent;             /* [n_entity, entity_dim]   */
    float *g_rel;             /* [n_relation, relation_dim] */
    float *g_modulus;         /* [1] (pRotatE) or null    */
    int32_t rows_clear;       /* mkb_pool_step / _bwd only, a promise of the caller: the g_ent rows of this batch's entities
                                 (pool ids, heads, tails) are all-zero on entry (fresh buffers; or the row-lazy optimizer's
                                 advance launch has just consumed and cleared them).  Rows that a single workgroup writes are
                                 then stored instead of read-modify-written.  0 = accumulate as always. */
} mkb_grads_t;

int mkb_abi_version(void);
const char *mkb_last_error(void);

/* Range check of the ids a scoring call is about to use: the reference's index_select (models/base.py:166-207) raises
 * IndexError for an id outside the table; the kernels index the tables directly, so the glue runs this (one small launch)
 * on user-supplied ids and raises when it next synchronises.  sample [B, 3], cand [n_cand] (or null); flag: one int32 on
 * the device, OR-ed with 1 (entity id in sample), 2 (relation id), 4 (candidate id) when something is out of range. */
int mkb_check_ids(const int64_t *sample, int64_t B, const int64_t *cand, int64_t n_cand, int64_t n_entity,
                  int64_t n_relation, int32_t *flag, void *stream);

/* ---- general scoring (arbitrary candidate ids) ------------------------------------------------------
 * mkb_score_fwd == model.forward(sample, negative_sample, mode)          (forward of each file under models/)
 *   sample [B,3]; cand [B,K] candidate entity ids (head-batch: heads, tail-batch: tails) or null with
 *   K = 1 for MKB_MODE_DEFAULT (candidate = the true tail, base.py:166-175); score [B,K] out.
 * mkb_score_bwd == autograd of the same call: accumulates d loss/d tables given dscore [B,K]
 *   (index_select backward = dense index_add_).
 */
int mkb_score_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *cand, int64_t B, int64_t K,
                  int mode, float *score, void *stream);
/* ws: caller-owned scratch of mkb_score_bwd_workspace_bytes(tb, B, K, mode) bytes, 256-byte aligned (the B*K (row, slot)
 * pairs radix-sorted by candidate, the queries, the sort's own storage); may be null when that returns 0 (default mode,
 * K = 1).  Nothing is allocated inside the call. */
int64_t mkb_score_bwd_workspace_bytes(const mkb_tables_t *tb, int64_t B, int64_t K, int mode);
int mkb_score_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *cand,
                  int64_t B, int64_t K, int mode, const float *dscore, void *ws, void *stream);

/* ---- self-adversarial loss ---------------------------------------------------------------------------
 * mkb_adversarial == losses.Adversarial(alpha)(positive_score, negative_score, weight) AND its gradient
 * (losses/adversarial.py:21-30): loss[1], dpos[B], dneg[B,K] out.  cnt (uint16 [B,K]) is an optional
 * per-column multiplicity (null = all ones) used by the pooled path, where a column is a pool position.
 * scratch: B+1 floats of caller-owned device memory (W and the per-row partial sums; the reduction is a
 * fixed tree, so the loss is bit-reproducible run to run).
 * weight_sum: null, or a device scalar holding W when the rows are a SHARD of a larger batch (data-parallel
 * ranks pass the all-reduced sum of weights; loss then is this shard's share of the global loss).
 */
int mkb_adversarial(const float *pos, const float *neg, const float *weight, const uint16_t *cnt, int64_t B,
                    int64_t K, float alpha, const float *weight_sum, float *loss, float *dpos, float *dneg,
                    float *scratch, void *stream);

/* ---- negative sampler --------------------------------------------------------------------------------
 * mkb_sampler_create == sampling.NegativeSampling.__init__ (negative_sampling.py:133-151): takes the two
 *   filter dictionaries of positive_triples() (negative_sampling.py:7-28) as CSR on the HOST:
 *   keys sorted ascending (head filter: r*n_entity+t -> heads; tail filter: h*n_relation+r -> tails),
 *   offsets [nk+1], values sorted ascending within each set.  Copies them to the device.  K <= 1024.
 * mkb_sampler_generate == NegativeSampling.generate(sample, mode) (negative_sampling.py:158-201), bit-exact
 *   with numpy's legacy MT19937 randint + np.in1d(assume_unique=True, invert=True) of numpy >= 1.24:
 *   neg [B,K] int64 out; optional outs for the pooled scoring path: pool [2K] int64 (the shared candidate
 *   draw), pos [B,K] int32 (neg[i,j] == pool[pos[i,j]]), cnt [B,2K] uint16 (multiplicity of each pool
 *   position in row i), touched [2K + 2B] int64 (the entity rows a training step on this batch reads: the pool,
 *   then the batch's heads, then its tails -- the id list of mkb_adam_rows_catchup / _step).  status [1] int32 device out: 0 / MKB_ERR_KEY / MKB_ERR_EMPTY (first failing row
 *   in status[1]); checked lazily by the host with mkb_sampler_status.
 */
typedef struct mkb_sampler mkb_sampler_t;
int mkb_sampler_create(mkb_sampler_t **out, int64_t n_entity, int64_t n_relation, int64_t K, uint32_t seed,
                       const int64_t *head_keys_host, int64_t n_head_keys, const int64_t *head_offsets_host,
                       const int64_t *head_values_host, const int64_t *tail_keys_host, int64_t n_tail_keys,
                       const int64_t *tail_offsets_host, const int64_t *tail_values_host, void *stream);
int mkb_sampler_generate(mkb_sampler_t *s, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                         int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream);
int mkb_sampler_status(mkb_sampler_t *s, void *stream); /* synchronises `stream`; returns 0 or the error */
/* Optional NON-PARITY pool draw (the reference has no counterpart; BASELINE north_star names it): kind 1 = rocRAND
 * Philox4x32-10 inside the same draw kernel (masked rejection like numpy's randint; counter based, so `draws` -- the number
 * of pools drawn so far -- is the whole generator state); kind 0 (default) = numpy's legacy MT19937, bit-exact negatives. */
int mkb_sampler_set_rng(mkb_sampler_t *s, int kind, uint64_t seed, uint64_t draws);
int mkb_sampler_get_rng(mkb_sampler_t *s, int *kind, uint64_t *seed, uint64_t *draws);
int mkb_sampler_get_state(mkb_sampler_t *s, uint32_t *key624_host, int32_t *pos_host, void *stream);
int mkb_sampler_set_state(mkb_sampler_t *s, const uint32_t *key624_host, int32_t pos, void *stream);
void mkb_sampler_destroy(mkb_sampler_t *s);

/* ---- pooled training step ----------------------------------------------------------------------------
 * == compose/pipeline.py:211-236 for one batch whose negatives come from ONE shared pool
 * (negative_sampling.py:166): positive forward (mode None), negative forward (mode), Adversarial,
 * backward into the dense gradient buffers.  Uses pool/cnt from mkb_sampler_generate.
 *   ws: workspace of mkb_pool_step_workspace_bytes() bytes; pos_score [B] out, pool_score [B,2K] out
 *   (score of row i against pool position p, meaningful where cnt > 0: elsewhere the tile kernels write 0 and the GEMM
 *   route of ComplEx / DistMult leaves the unmasked product or, beyond the deepest position the row's tile uses, nothing),
 *   loss [1] out; weight_sum as in
 *   mkb_adversarial (null = sum of this call's weights).
 */
int mkb_pool_supported(const mkb_tables_t *tb, int64_t B, int64_t K); /* 1 if the pooled kernels cover this shape */
int64_t mkb_pool_step_workspace_bytes(const mkb_tables_t *tb, int64_t B, int64_t K);
int mkb_pool_step(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                  const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                  const float *weight_sum, float *pos_score, float *pool_score, float *loss, void *ws, void *stream);
/* The two halves of mkb_pool_step (mkb_pool_step == fwd then bwd on the same buffers).  A caller that shards the
 * embedding DIMENSIONS over devices sums pos_score / pool_score across devices between the halves (scores are sums
 * over dims; gamma must enter once: e.g. only one device's tables carry it) and each device then back-propagates into its
 * own slice with no further exchange.
 * The backward half continues the forward half's workspace: besides the queries it holds the per-step occurrence counts
 * of the batch's entities (a gradient row whose entity occurs once is written without atomics) -- same ws, same batch,
 * forward first. */
int mkb_pool_step_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt, int64_t B,
                      int64_t K, int mode, float *pos_score, float *pool_score, void *ws, void *stream);
int mkb_pool_step_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const float *weight,
                      const int64_t *pool, const uint16_t *cnt, int64_t B, int64_t K, int mode, float alpha,
                      const float *weight_sum, const float *pos_score, const float *pool_score, float *loss, void *ws,
                      void *stream);

/* pooled model.forward / its autograd as separate calls (README-style loops that call the model and the loss
 * themselves): pool_score [B,2K] out; dpool_score [B,2K] = d loss / d pool_score in. */
int mkb_pool_score_fwd(const mkb_tables_t *tb, const int64_t *sample, const int64_t *pool, const uint16_t *cnt,
                       int64_t B, int64_t K, int mode, float *pool_score, void *ws, void *stream);
int mkb_pool_score_bwd(const mkb_tables_t *tb, const mkb_grads_t *gr, const int64_t *sample, const int64_t *pool,
                       const uint16_t *cnt, int64_t B, int64_t K, int mode, const float *dpool_score, void *ws,
                       void *stream);

/* ---- distillation loss --------------------------------------------------------------------------------
 * == losses.KlDivergence()(student_score, teacher_score, T) (losses/kl_divergence.py:22-29), called by
 * distillation.Distillation.distill (distillation/distillation.py:633-683) from KdmkbModel.forward
 * (distillation/kdmkb_model.py:337-349): mean over all n*m entries of t (log t - log p), t = softmax(teacher / T, dim=1),
 * p = softmax(student / T, dim=1).  student, teacher: [n, m] row-major.  loss: 1 float.  dstudent [n, m] = d loss / d
 * student; dteacher [n, m] = d loss / d teacher, or null (the reference scores the teacher under no_grad).
 * scratch: n floats.
 */
int mkb_kl_divergence(const float *student, const float *teacher, int64_t n, int64_t m, float T, float *loss,
                      float *dstudent, float *dteacher, float *scratch, void *stream);

/* ---- dense Adam --------------------------------------------------------------------------------------
 * == torch.optim.Adam(lr, betas, eps).step() + zero_grad() for one parameter tensor as the README loop
 * uses it (README.md:123-126, pipeline.py:238-240): every element moves every step.  step is 1-based.
 * zero_grad != 0 also clears g (fused optimizer.zero_grad()).
 */
int mkb_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, float lr,
                  float beta1, float beta2, float eps, int zero_grad, void *stream);

/* Row-lazy form of the same dense Adam (identical arithmetic, see mkb_amd/csrc/adam.hip): `last` [n_rows] int32
 * (zero-initialised) holds the step each row is current through, `consts` [capacity, 2] float holds the per-step
 * scalars recorded by mkb_adam_rows_step.
 *   mkb_adam_rows_catchup: replay the pending zero-gradient steps of the rows in ids (null = every row, i.e. a
 *     flush) so that they are current through step_upto;
 *   mkb_adam_rows_step: apply step `step` with the real gradient to the rows in ids (duplicates allowed; they must
 *     be current through step-1) and clear their gradient rows.
 */
/* draw_ahead: null, or a sampler whose NEXT pool draw (the single-workgroup MT19937 kernel every mkb_sampler_generate
 * starts with) runs as one more workgroup of the catch-up launch (ids != null); the next mkb_sampler_generate on that
 * sampler then only filters.  Same stream as the sampler's other calls; the negatives are bit-identical to drawing at
 * generate time, and mkb_sampler_get_state keeps reporting the state before the pool drawn ahead. */
int mkb_adam_rows_catchup(float *param, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts, int64_t n_rows,
                          int64_t D, const int64_t *ids, int64_t n_ids, int64_t step_upto, float beta1, float beta2,
                          float eps, mkb_sampler_t *draw_ahead, void *stream);
/* mkb_sampler_generate(sampler, sample, B, mode, neg, pool, pos, cnt, touched) and mkb_adam_rows_catchup over the rows
 * that batch touches (its pool, heads and tails) as ONE launch that also draws the sampler's next pool: the whole
 * sampler runs in the shadow of the optimizer's catch-up.  Same outputs, bit for bit, as the two calls (size <= 512). */
int mkb_adam_rows_catchup_generate(float *param, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                   int64_t n_rows, int64_t D, int64_t step_upto, float beta1, float beta2, float eps,
                                   mkb_sampler_t *sampler, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                                   int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream);
typedef struct {
    float *param, *grad, *exp_avg, *exp_avg_sq; /* a small dense tensor (e.g. the relation table), 16-byte aligned */
    int64_t n;                                  /* elements */
    int64_t step;                               /* its own step count (>= 1) */
} mkb_adam_dense_t;
/* mkb_adam_step for up to 8 parameter tensors in ONE launch (each with its own step count): what an optimizer over a model's
 * few dense tensors does per step -- small models are bound by launches, not by bytes (Umls TransE-64: 9 launches of ~5 us).
 * draw_ahead: as in mkb_adam_rows_catchup (the sampler's next pool draw as one more workgroup of this launch), or null. */
int mkb_adam_step_multi(const mkb_adam_dense_t *tensors_host, int n_tensors, float lr, float beta1, float beta2, float eps,
                        int zero_grad, mkb_sampler_t *draw_ahead, void *stream);
/* rider: null, or one dense tensor that takes its mkb_adam_step (with zero_grad) inside the same launch. */
int mkb_adam_rows_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                       int64_t n_rows, int64_t D, const int64_t *ids, int64_t n_ids, int64_t step, float lr, float beta1,
                       float beta2, float eps, const mkb_adam_dense_t *rider, void *stream);
/* "Advance" form of mkb_adam_rows_catchup / _catchup_generate: the real step of the touched rows is deferred too.  The rows
 * of step t's batch were made current through t-1 before its forward pass, so after backward their state is "current
 * through t-1, gradient of step t in the gradient row" -- and stays that way until the row is next read: the replay takes
 * the row's gradient row for its FIRST pending step (zero for rows that were not touched then, which is the
 * zero-gradient step bit for bit) and clears it.  mkb_adam_rows_step is then only needed for the very first step
 * (`last` = 0 means "never touched").  grad: the table's dense gradient [n_rows, D]; lr: learning rate of step
 * step_upto (recorded into consts[step_upto] by the launch); rider: the small dense tensor that takes ITS step
 * (rider->step) in the same launch, or null.  Callers must bring a row current (catch-up / advance) BEFORE a backward
 * pass writes its gradient row, and route every read of the tables through an advance with ids = null (flush). */
int mkb_adam_rows_advance(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                          int64_t n_rows, int64_t D, const int64_t *ids, int64_t n_ids, int64_t step_upto, float lr,
                          float beta1, float beta2, float eps, const mkb_adam_dense_t *rider, mkb_sampler_t *draw_ahead,
                          void *stream);
int mkb_adam_rows_advance_generate(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                   int64_t n_rows, int64_t D, int64_t step_upto, float lr, float beta1, float beta2, float eps,
                                   const mkb_adam_dense_t *rider, mkb_sampler_t *sampler, const int64_t *sample, int64_t B,
                                   int mode, int64_t *neg, int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched,
                                   void *stream);

/* ---- row-sharded entity table (several GPUs; the reference is single-process: no counterpart to cite) ----------------
 * Rank g of `world` owns the entity rows e with e % world == g, at shard index e / world (table, dense gradient and Adam
 * state alike).  mkb_amd/table_rows.py moves rows between owners and users with these three calls and torch.distributed
 * (RCCL) collectives; the scoring step itself runs unchanged on a compact table [pool rows | ... | heads | tails].
 *
 * mkb_rows_route: group row requests by owner, request order kept inside a group.  sample_layout != 0: ids = sample [n, 3],
 *   the requests are its n heads followed by its n tails (2n requests); sample_layout == 0: ids = a flat list of n global
 *   entity ids.  send_ids = shard indices in grouped order (the payload of the id all-to-all), slot = where request j went,
 *   counts [world] = requests per owner (the all-to-all's split sizes), compact [n, 3] (sample layout only, or null) = the
 *   triples re-addressed into the compact table: (row0 + slot[j], r, row0 + slot[n + j]).
 * A row segment lists rows of the shard: world == 0: ids are shard indices; world > 0: ids are GLOBAL entity ids and only
 *   those this rank owns are touched (gather: the others become zero rows, so that an all-reduce over the ranks completes
 *   the block; scatter: skipped).  local_ids (gather, optional out [n]): the shard index of each entry, or -1.
 * mkb_rows_gather: rows[j] = shard[ids[j]] for up to 4 segments in one launch; riders of the same launch: weight_sum[0] =
 *   sum of weight [n_weight] (fixed-order tree; null = none) and clearing `zero` (zero_bytes, whole floats, 16-byte aligned; 0 = none).
 * mkb_rows_scatter_add: grad[ids[j]] += rows[j] (fp32 atomics: duplicates add); riders: dense_dst [dense_n] += dense_src,
 *   copy_dst [copy_n <= 256] = copy_src (how the step's loss leaves the reused step buffers without a launch of its own).
 * occ (optional, [n_local] uint32, zero-initialised ONCE by the caller): the gather launch counts how often each shard row
 *   is listed in its segments; the scatter launch of the SAME segments adds rows listed once without atomics and resets the
 *   counts.  null = always atomics.
 * bad (optional, int32 [1], zeroed by the caller): ids the reference's gather would answer with IndexError
 *   (models/base.py:193-207) never touch memory here -- bit 0: a negative id reached mkb_rows_route (routed as id 0);
 *   bit 1: a shard index outside [0, n_local) reached the gather / scatter (row skipped: zeros / nothing added). */
typedef struct {
    const int64_t *ids;
    int64_t n;
    float *rows;          /* [n, D] */
    int32_t world, rank;
    int64_t *local_ids;   /* gather only; may be null */
} mkb_row_seg_t;
int mkb_rows_route(const int64_t *ids, int64_t n, int sample_layout, int world, int64_t row0, int64_t *send_ids,
                   int32_t *slot, int64_t *counts, int64_t *compact, int32_t *bad, void *stream);
int mkb_rows_gather(const float *shard, int64_t n_local, int64_t D, const mkb_row_seg_t *segs, int n_segs,
                    const float *weight, int64_t n_weight, float *weight_sum, void *zero, int64_t zero_bytes, uint32_t *occ,
                    int32_t *bad, void *stream);
int mkb_rows_scatter_add(float *grad, int64_t n_local, int64_t D, const mkb_row_seg_t *segs, int n_segs, float *dense_dst,
                         const float *dense_src, int64_t dense_n, float *copy_dst, const float *copy_src, int64_t copy_n,
                         uint32_t *occ, int32_t *bad, void *stream);
/* mkb_adam_rows_advance (grad != null) / mkb_adam_rows_catchup (grad == null) for a shard of such a table: the rows to visit
 * are global_ids [n_global] (entries other ranks own are skipped) followed by local_ids [n_local_ids] (shard indices).
 * Negative entries of any id list of the row-lazy calls are skipped. */
int mkb_adam_rows_advance_sharded(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last, float *consts,
                                  int64_t n_rows, int64_t D, const int64_t *global_ids, int64_t n_global, int world, int rank,
                                  const int64_t *local_ids, int64_t n_local_ids, int64_t step_upto, float lr, float beta1,
                                  float beta2, float eps, const mkb_adam_dense_t *rider, mkb_sampler_t *draw_ahead,
                                  void *stream);

/* ... and with this rank's mkb_sampler_generate riding the same launch (the pool ids are the sampler's own): the sharded
 * counterpart of mkb_adam_rows_advance_generate / _catchup_generate (grad == null). */
int mkb_adam_rows_advance_sharded_generate(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *last,
                                           float *consts, int64_t n_rows, int64_t D, int world, int rank,
                                           const int64_t *local_ids, int64_t n_local_ids, int64_t step_upto, float lr,
                                           float beta1, float beta2, float eps, const mkb_adam_dense_t *rider,
                                           mkb_sampler_t *sampler, const int64_t *sample, int64_t B, int mode, int64_t *neg,
                                           int64_t *pool, int32_t *pos, uint16_t *cnt, int64_t *touched, void *stream);

/* ---- the row-sharded step's collectives, issued by the library (no reference counterpart: mkb is single-process) ------------
 * mkb_amd/table_rows.py used to issue the step's six collectives through torch.distributed and to read the all-to-alls' split
 * sizes back with an event wait: measured host-bound (0.43 ms / step against 0.19 ms of kernels, round 4).  With these entry
 * points the library holds its own RCCL communicators (bound at run time: mkb_rows_comm_available() is 0 on a box without
 * librccl, everything else still loads) and a step's communication is plan (ahead, side stream) + take + 2 x exchange.
 *
 * mkb_rows_comm_unique_id: rank 0 fills id_host [MKB_ROWS_COMM_ID_BYTES]; the caller hands the blob to every rank by whatever
 *   means it has (mkb_amd: one torch.distributed broadcast at set-up).
 * mkb_rows_comm_create: collective over the `world` ranks (each on its own current device).  max_requests = the largest 2 x b
 *   (positive-row requests of one rank's batch) any rank will ever plan: the id lists travel in fixed blocks of that capacity.
 * mkb_rows_comm_plan (slot in [0, 4): the caller's ring of plans in flight): on side_stream (after the work queued on
 *   after_stream so far, when given) -- mkb_rows_route on sample [b, 3] (send_ids [2b], slot_of [2b], counts [world],
 *   compact [b, 3]: as there), then the id exchange with in-band counts, then `want` [want_cap >= world's total requests to this
 *   owner; 2 b x world always suffices] = the shard indices the ranks ask this owner for, requester after requester.  Nothing
 *   returns to the host except through the mailbox below.  bad: as mkb_rows_route; bit 2 = a peer's block was malformed.
 * mkb_rows_comm_take: the split sizes of that plan -- sent_host [world] rows this rank asks each owner for, wanted_host [world]
 *   rows each rank asks this owner for -- read from a host-coherent mailbox the plan's last kernel wrote (no HIP call; spins
 *   only when the plan has not executed yet -- for at most MKB_ROWS_TAKE_TIMEOUT_S seconds (default 120), then MKB_ERR_HIP), and
 *   `stream` is made to wait for the plan.
 * mkb_rows_comm_exchange: on `stream`: the all-reduce (sum, in place) of reduce [reduce_n] floats (0 = none), then ONE RCCL
 *   group with the all-to-all of rows of D floats (MKB_ROWS_ONE_GROUP=1, and always at world 1: both in one group): send_rows_host[p] rows to rank p from `send` (consecutive), recv_rows_host[p] rows from
 *   rank p into `recv` (null count vectors = no all-to-all).  Forward: owners send `wanted`, users receive `sent`; the
 *   gradients' way back swaps the two.
 * mkb_rows_comm_stats: plans made, takes that found their plan not executed yet, and how many of those found `stream` idle
 *   (only those are bubbles on the device: the others mean the host ran ahead of it). */
#define MKB_ROWS_COMM_ID_BYTES 256
#define MKB_ROWS_MAX_WORLD 64
typedef struct mkb_rows_comm mkb_rows_comm_t;
/* what a plan posts for the host (inside a communicator: host-coherent memory): seq is stored LAST, with system-scope release */
typedef struct {
    int64_t seq;
    int64_t sent[MKB_ROWS_MAX_WORLD];    /* rows this rank asks each owner for (= the route's counts) */
    int64_t wanted[MKB_ROWS_MAX_WORLD];  /* rows each rank asks this owner for                        */
} mkb_rows_mailbox_t;
/* The two kernels of mkb_rows_comm_plan on their own (a caller with a transport of its own moves the blocks between them; the
 * tests play several ranks in one process with them).  blocks: [world][1 + cap] int64 -- pack writes block w = [counts[w] | the
 * ids of owner w's group of send_ids]; unpack reads block j = what rank j asks this owner for, writes `want` (requester after
 * requester), then mail->sent = counts, mail->wanted, and mail->seq = seq last.  mail: any device-visible memory. */
int mkb_rows_blocks_pack(const int64_t *counts, const int64_t *send_ids, int64_t *blocks, int world, int64_t cap, void *stream);
int mkb_rows_blocks_unpack(const int64_t *blocks, const int64_t *counts, int64_t *want, int64_t want_cap, mkb_rows_mailbox_t *mail,
                           int64_t seq, int32_t *bad, int world, int64_t cap, void *stream);
int mkb_rows_comm_available(void);
int mkb_rows_comm_unique_id(uint8_t *id_host);
/* In-process transport (ABI 6): RCCL refuses two ranks on one device, so plan / take / exchange cannot be driven for world > 1 on a
 * one-GPU box through it.  A communicator made by mkb_rows_comm_create_loopback moves its blocks and rows through a hub in THIS
 * process instead (device-to-device copies ordered by events, contributions of the all-reduce added in rank order): every rank is a
 * host thread with its own streams, all on the current device, and all the entry points above behave as they do over RCCL --
 * except that a peer that never makes the matching call, or disagrees about a split size, produces an ERROR after
 * MKB_ROWS_LOOP_TIMEOUT_S (default 20) seconds where RCCL would hang.  tests/test_gpu_rows_loopback.py; not used by the product. */
int mkb_rows_loop_hub_create(int world, void **hub);
void mkb_rows_loop_hub_destroy(void *hub);  /* after the communicators made over it */
int mkb_rows_comm_create_loopback(void *hub, int rank, int64_t max_requests, mkb_rows_comm_t **out);
int mkb_rows_comm_create(const uint8_t *id_host, int rank, int world, int64_t max_requests, mkb_rows_comm_t **out);
void mkb_rows_comm_destroy(mkb_rows_comm_t *comm);
int mkb_rows_comm_plan(mkb_rows_comm_t *comm, int slot, const int64_t *sample, int64_t b, int64_t row0, int64_t *send_ids,
                       int32_t *slot_of, int64_t *counts, int64_t *compact, int64_t *want, int64_t want_cap, int32_t *bad,
                       void *after_stream, void *side_stream);
int mkb_rows_comm_take(mkb_rows_comm_t *comm, int slot, int64_t *sent_host, int64_t *wanted_host, void *stream);
int mkb_rows_comm_exchange(mkb_rows_comm_t *comm, float *reduce, int64_t reduce_n, const float *send,
                           const int64_t *send_rows_host, float *recv, const int64_t *recv_rows_host, int64_t D, void *stream);
int mkb_rows_comm_stats(mkb_rows_comm_t *comm, int64_t *plans, int64_t *takes_that_waited, int64_t *waited_with_idle_stream);

/* ---- filtered ranking --------------------------------------------------------------------------------
 * == evaluation.Evaluation.compute_score for head-/tail-batch (evaluation/evaluation.py:217-279) with the
 * candidate list and filter bias of datasets.base.TestDataset (datasets/base.py:196-241): for each test triple
 * the rank of the target among all n_entity candidates, other true triples biased by -100000.
 *   true_keys: ascending int64 keys of ALL true triples (device), ordered for the mode so that one query's
 *   filter set is a contiguous range: tail-batch (h*n_relation + r)*n_entity + t, head-batch
 *   (t*n_relation + r)*n_entity + h.  rank [B] int64 out (1-based; ties count in the target's favour).
 *   ws: mkb_rank_workspace_bytes(tb, B) bytes, 256-byte aligned (queries + the [B, n_entity] score block).
 */
int64_t mkb_rank_workspace_bytes(const mkb_tables_t *tb, int64_t B);
int mkb_rank(const mkb_tables_t *tb, const int64_t *sample, int64_t B, int mode, const int64_t *true_keys,
             int64_t n_true, int64_t *rank, void *ws, int64_t ws_bytes, void *stream);
/* mkb_rank, and the score block the ranks were counted on handed out as well: scores [B, n_entity] fp32 =
 * model(sample, negative_sample = every entity id in order, mode) of evaluation.py:237 BEFORE the filter bias is added
 * (the same launches as mkb_rank plus one copy; the parity tests compare this block with the oracle's at full size, and
 * utils-style "score against all entities" callers get it without a [B, N, D] gather).  ABI 6. */
int mkb_rank_scores(const mkb_tables_t *tb, const int64_t *sample, int64_t B, int mode, const int64_t *true_keys,
                    int64_t n_true, int64_t *rank, float *scores, void *ws, int64_t ws_bytes, void *stream);

/* ---- per-kernel timing (measurement aid, no reference counterpart) -------------------------------------
 * When enabled, the launches of the named kernel class are bracketed by hipEvents recorded on the SAME stream
 * the kernel is launched on (on = N > 1: every N-th launch only -- the two event records cost ~6 us of stream time
 * each, which a sampled measurement keeps out of most steps).  mkb_profile_read synchronises, returns the number of bracketed launches and
 * their summed duration in milliseconds, and resets the counters.  kernel: 0 = pooled backward (the dq and dx passes
 * in one launch; for ComplEx / DistMult the dQ GEMM),
 * 1 = pooled forward, 2 = Adam, 3 = sampler (draw + filter), 4 = adversarial loss, 5 = general forward,
 * 6 = general backward, 7 = pooled backward dx pass when it is launched alone (the GEMM route, MKB_POOL_SPLIT_BWD).  At most 8192 launches are kept between reads.
 */
#define MKB_PROF_POOL_BWD_Q 0
#define MKB_PROF_POOL_FWD 1
#define MKB_PROF_ADAM 2
#define MKB_PROF_SAMPLER 3
#define MKB_PROF_LOSS 4
#define MKB_PROF_GENERAL_FWD 5
#define MKB_PROF_GENERAL_BWD 6
#define MKB_PROF_POOL_BWD_X 7
#define MKB_PROF_KINDS 8
int mkb_profile_enable(int kernel, int on);
/* measurement aid: the shader clock in MHz as one wave sees it over ~20 us (s_memtime cycles per 100 MHz s_memrealtime tick) */
int mkb_debug_sclk_mhz(float *out_mhz_device, void *stream);
int mkb_profile_read(int kernel, int64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MKB_HIP_H */

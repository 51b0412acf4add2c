"""Multi-GPU training: batch-row data parallelism with a SPARSE gradient exchange over RCCL (xGMI).

The reference has no distributed code at all; this is the MI355X-native scale-out of
compose/pipeline.py's step.  One process per GPU (``torch.distributed``, backend ``"nccl"`` = RCCL):

* tables, optimizer state and the sampler's MT19937 state are REPLICATED; every rank draws the identical
  candidate pool each step (negative_sampling.py:166 semantics for the GLOBAL batch) and filters / scores only
  its own rows -- the negatives of the global batch are bit-identical to a single-GPU run over the same rows;
* the loss normaliser ``W = sum(weight)`` is the global one (all-reduced scalar, adversarial.py:28-29), so the
  summed rank gradients equal the gradient of the global batch;
* a step touches few table rows (the <= 2K pool rows + this rank's heads/tails + <= n_relation relation rows),
  so instead of all-reducing the dense 116 MB gradient the ranks all-gather the touched row ids (a few KB),
  form the identical sorted union, and all-reduce only those rows packed into one buffer (entity rows,
  then the relation table, then scalars).  When the union covers most of the table (many ranks, small table)
  the dense all-reduce is used instead -- whichever moves fewer bytes;
* every rank then applies the identical dense optimizer step to its replica.

xGMI is a point-to-point mesh; the single packed all-reduce per step keeps the collective count at one
latency-bound launch plus two tiny ones (ids all-gather, W all-reduce), which is what matters at these sizes
(tens of MB at most).  Everything here is device-agnostic torch code and is exercised on CPU with the ``gloo``
backend in tests/test_parallel_gloo.py.
"""
import torch
import torch.distributed as dist

from . import _links

__all__ = ["DimShardedStep", "SparseGradExchange", "allreduce_touched_rows", "gather_dims", "shard_dims", "shard_rows"]


def shard_rows(n_rows, rank, world):
    """Contiguous row range [lo, hi) of a global batch owned by ``rank``."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def allreduce_touched_rows(grad, local_ids, extras=(), group=None, dense_threshold=0.6, equal_counts=False):
    """Sum ``grad`` ([N, D], dense, zero outside the rows this rank touched) across ranks, moving only the rows
    that some rank touched.  ``local_ids``: 1-D int64 ids this rank touched (duplicates allowed).  ``extras``:
    small tensors all-reduced in the same collective (relation table gradient, scalars).  In place.
    ``equal_counts``: every rank passes the same number of ids (skips one tiny collective + host sync).
    Returns the number of entity rows moved (N for the dense fallback)."""
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    n, d = grad.shape
    # 1) identical union of touched ids on every rank (pad to a common length with -1)
    if equal_counts:
        padded = local_ids.contiguous()
    else:
        count = torch.tensor([local_ids.numel()], device=grad.device, dtype=torch.int64)
        counts = [torch.zeros_like(count) for _ in range(world)]
        dist.all_gather(counts, count, group=group)
        cap = int(max(c.item() for c in counts))
        padded = torch.full((cap,), -1, device=grad.device, dtype=torch.int64)
        padded[: local_ids.numel()] = local_ids
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    union = torch.unique(torch.cat(gathered))
    union = union[union >= 0]
    allreduce_touched_rows.last_union = union
    extras = [e for e in extras if e is not None]
    flat_extra = [e.reshape(-1) for e in extras]
    if union.numel() >= dense_threshold * n:
        buf = torch.cat([grad.reshape(-1)] + flat_extra)
        dist.all_reduce(buf, group=group)
        grad.copy_(buf[: n * d].view(n, d))
        moved = n
    else:
        buf = torch.cat([grad.index_select(0, union).reshape(-1)] + flat_extra)
        dist.all_reduce(buf, group=group)
        grad.index_copy_(0, union, buf[: union.numel() * d].view(-1, d))
        moved = int(union.numel())
    off = moved * d
    for e in extras:
        e.copy_(buf[off: off + e.numel()].view_as(e))
        off += e.numel()
    return moved


class SparseGradExchange:
    """Per-step collective part of the data-parallel training step for one ``mkb_amd`` model.

    ``W = ex.weight_sum(weight)``   before the fused step (pass as ``weight_sum=``),
    ``ex(sample, negative_sample)`` after it: ``model.*.grad`` then hold the global-batch gradient on every rank.
    ``ex.loss(local_loss)``         global loss (sum of the ranks' shares), for logging.
    """

    def __init__(self, model, group=None, equal_batches=False):
        self.model, self.group, self.equal_batches = model, group, equal_batches
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.last_rows_moved = 0
        opt = _links.owner(model.entity_embedding)
        if opt is not None and hasattr(opt, "stop_deferring"):
            # rows only OTHER ranks touched receive their gradient here without a catch-up before it: the optimizer's
            # deferred real step (which reads "gradient row = the row's first pending step") cannot be used
            opt.stop_deferring()

    def weight_sum(self, weight):
        w = weight.sum().reshape(1)
        if self.world > 1:
            dist.all_reduce(w, group=self.group)
        return w

    def __call__(self, sample, negative_sample, extra_scalars=()):
        if self.world == 1:
            return
        m = self.model
        info = getattr(negative_sample, "_mkb_pool", None)
        pool_ids = info.pool if info is not None else negative_sample.reshape(-1)
        ids = torch.cat([sample[:, 0], sample[:, 2], pool_ids])
        extras = [m.relation_embedding.grad]
        if m.name == "pRotatE":
            extras.append(m.modulus.grad)
        extras.extend(extra_scalars)
        self.last_rows_moved = allreduce_touched_rows(m.entity_embedding.grad, ids, extras, self.group,
                                                      equal_counts=self.equal_batches)
        if _links.owner(m.entity_embedding) is not None:
            # row-lazy Adam must step every row ANY rank touched: hand it the gathered id list
            _links.mark_touched(m.entity_embedding, self.last_union, replace=True)

    @property
    def last_union(self):
        return allreduce_touched_rows.last_union

    def loss(self, local_loss):
        t = local_loss.detach().reshape(1).clone()
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return t.reshape(())


# =====================================================================================================
# Dimension sharding ("tensor parallel" over the embedding dimension)
#
# Every score of the five models is a SUM OVER EMBEDDING DIMS of terms that couple only same-index components of
# h, r, t (complex models: the same complex index).  So the tables can be split by COLUMNS: rank g keeps
# ent[:, dims_g] / rel[:, dims_g] (1/G of the memory), scores every row of the global batch on its dims, and the
# only exchange per step is ONE all-reduce of the partial scores [B, 2K+1] (2 MB at B = 1024) -- after which loss,
# backward and Adam are entirely local to the rank's columns: no gradient exchange at all, no replicated optimizer
# work, table and optimizer state sharded G ways.  Compare with row/batch sharding, where the ranks must exchange
# the touched gradient ROWS (tens of MB per step, see SparseGradExchange above).
# Weak scaling: the global batch grows with G (G x 1024 rows), each rank does all rows x 1/G of the dims.


def _dim_slices(model, rank, world):
    """(entity column index, relation column index) of this rank's dims, matching the [real | imag] row layout."""
    d = model.hidden_dim
    if d < world:
        raise ValueError(f"hidden_dim {d} is smaller than the world size {world}")
    # Uneven splits are fine, but every rank should get a multiple of 4 (or 2) dims so that its kernels can use
    # vector lanes and packed math: 1000 dims over 8 ranks -> 128 x 4 + 124 x 4, not 8 x 125.
    unit = 4 if d % 4 == 0 and d // 4 >= world else (2 if d % 2 == 0 and d // 2 >= world else 1)
    own = torch.tensor_split(torch.arange(d // unit), world)[rank]
    own = (own[:, None] * unit + torch.arange(unit)[None, :]).reshape(-1)
    if model.name == "RotatE":
        return torch.cat([own, d + own]), own
    if model.name == "ComplEx":
        return torch.cat([own, d + own]), torch.cat([own, d + own])
    return own, own


def shard_dims(model, rank, world, device=None):
    """A model of the same class holding this rank's 1/world of the embedding dims of ``model`` (a full model,
    e.g. freshly initialised with the reference's seed on the CPU).  Its forward returns PARTIAL scores (rank 0's include
    gamma, so that the all-reduced sum holds it once)."""
    import math

    ec, rc = _dim_slices(model, rank, world)
    local = model.__class__(hidden_dim=int(rc.numel() // (2 if model.name == "ComplEx" else 1)), entities={v: k for k, v in model.entities.items()},
                            relations={v: k for k, v in model.relations.items()}, gamma=model.gamma.item())
    with torch.no_grad():
        local.entity_embedding.copy_(model.entity_embedding.detach().cpu()[:, ec])
        local.relation_embedding.copy_(model.relation_embedding.detach().cpu()[:, rc])
        local.embedding_range.copy_(model.embedding_range.detach().cpu())
        if hasattr(model, "modulus"):
            local.modulus.copy_(model.modulus.detach().cpu())
    phase_div = torch.tensor(model.embedding_range.item() / math.pi, dtype=torch.float32).item()
    uses_gamma = model.name in ("TransE", "RotatE", "pRotatE")
    # partial scores: rank 0's carry gamma, the others' do not, so the cross-rank sum holds it exactly once (no add-gamma
    # kernel behind the all-reduce)
    local._consts_override = (model.gamma.item() if (uses_gamma and rank == 0) else 0.0, phase_div)
    local._dim_shard = (rank, world, model.gamma.item(), uses_gamma)
    return local if device is None else local.to(device)


def gather_dims(local, group=None):
    """Reassemble the full tables from every rank's dimension shard: (entity_embedding, relation_embedding)."""
    rank, world, _, _ = local._dim_shard
    opt = _links.owner(local.entity_embedding)
    if opt is not None:
        opt.flush(local.entity_embedding)
    outs = []
    for p, complex_rows in ((local.entity_embedding, local.name in ("RotatE", "ComplEx")),
                            (local.relation_embedding, local.name == "ComplEx")):
        width = torch.tensor([p.shape[1]], device=p.device)
        widths = [torch.zeros_like(width) for _ in range(world)]
        dist.all_gather(widths, width, group=group)
        widths = [int(w.item()) for w in widths]
        wmax = max(widths)
        mine = torch.zeros(p.shape[0], wmax, dtype=p.dtype, device=p.device)
        mine[:, : p.shape[1]] = p.data
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        parts = [q[:, :w] for q, w in zip(parts, widths)]
        if complex_rows:
            outs.append(torch.cat([q[:, : q.shape[1] // 2] for q in parts] + [q[:, q.shape[1] // 2:] for q in parts], dim=1))
        else:
            outs.append(torch.cat(parts, dim=1))
    return outs[0], outs[1]


class DimShardedStep:
    """The fused training step (mkb_amd.fused.FusedTrainStep) over a dimension-sharded model: forward half on the
    local dims, ONE all-reduce of the partial scores, backward half on the local dims.  Every rank must be given
    the same (global) batch and draw the same negatives (same sampler seed).

    ``micro_batches`` (default 2 when there is more than one rank) splits the rows of the batch so that the
    all-reduce of one part's scores runs on RCCL's stream while the other part's forward / backward kernels run:
    [fwd 0][fwd 1 | all-reduce 0][bwd 0 | all-reduce 1][bwd 1].  The loss normaliser W is the whole batch's."""

    def __init__(self, local_model, alpha, group=None, micro_batches=None):
        from . import _hip
        from .fused import _workspace

        self._hip, self._workspace = _hip, _workspace
        self.model, self.alpha, self.group = local_model, float(alpha), group
        self.rank, self.world, self.gamma, self.uses_gamma = local_model._dim_shard
        self.micro_batches = (2 if self.world > 1 else 1) if micro_batches is None else micro_batches

    def sampled(self, sample, weight, sampler, mode):
        """``step(sample, weight, sampler.generate(sample, mode), mode)`` with the sampler folded into the row-lazy
        optimizer's catch-up launch (see ``FusedTrainStep.sampled``); every rank draws the same negatives."""
        ent = self.model.entity_embedding
        lazy = _links.owner(ent)
        sample = self._hip.contiguous(sample, torch.int64)
        if lazy is not None and sampler.size <= 512 and sample.is_cuda:
            neg = sampler.generate_with_catch_up(sample, mode, lazy, ent)
        else:
            neg = sampler.generate(sample=sample, mode=mode)
        self.negative_sample = neg
        return self(sample, weight, neg, mode)

    def __call__(self, sample, weight, negative_sample, mode):
        _hip, m = self._hip, self.model
        info = negative_sample._mkb_pool
        mode_id = _hip.mode_id(mode)
        sample = _hip.contiguous(sample, torch.int64)
        weight = _hip.contiguous(weight, torch.float32)
        B, K = sample.shape[0], info.size
        dev = sample.device
        params = [m.entity_embedding, m.relation_embedding] + ([m.modulus] if m.name == "pRotatE" else [])
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        gr = _hip.Grads(m.entity_embedding.grad.data_ptr(), m.relation_embedding.grad.data_ptr(),
                        m.modulus.grad.data_ptr() if m.name == "pRotatE" else None)
        ent = m.entity_embedding
        lazy = _links.owner(ent)
        if lazy is not None:
            ids = info.touched if info.touched is not None else torch.cat([info.pool, sample[:, 0], sample[:, 2]])
            done = lazy._state(ent).get("caught_up")
            if done is None or done[0] is not ids or done[1] != lazy._state(ent)["n"]:  # (sampled() already did it)
                lazy.catch_up(ent, ids)
            _links.mark_touched(ent, ids)
        lib, tb = _hip.lib(), m._tables()
        nmb = max(1, min(self.micro_batches, B // 8))
        bounds = [(B * i // nmb, B * (i + 1) // nmb) for i in range(nmb)]
        wsum = weight.sum().reshape(1) if nmb > 1 else None          # W of the WHOLE batch (every rank has all rows)
        parts, works = [], []
        with _hip.on_device(dev):
            for lo, hi in bounds:                                     # forward halves + their all-reduces (async)
                b = hi - lo
                scores = torch.empty(b * (2 * K + 1), dtype=torch.float32, device=dev)  # [pos | pool]: one collective
                pos, S = scores[:b], scores[b:].view(b, 2 * K)
                ws = self._workspace(m, b, K, slot=len(parts))
                _hip.check(lib.mkb_pool_step_fwd(tb, _hip.ptr(sample[lo:hi]), _hip.ptr(info.pool), _hip.ptr(info.cnt[lo:hi]),
                                                 b, K, mode_id, _hip.ptr(pos), _hip.ptr(S), _hip.ptr(ws),
                                                 _hip.stream_ptr()), "mkb_pool_step_fwd")
                works.append(dist.all_reduce(scores, group=self.group, async_op=True) if self.world > 1 else None)
                parts.append((lo, hi, scores, pos, S, ws))
            total = None
            for (lo, hi, scores, pos, S, ws), work in zip(parts, works):  # backward halves as their scores arrive
                b = hi - lo
                if work is not None:
                    work.wait()
                loss = torch.empty(1, dtype=torch.float32, device=dev)
                _hip.check(lib.mkb_pool_step_bwd(tb, gr, _hip.ptr(sample[lo:hi]), _hip.ptr(weight[lo:hi]), _hip.ptr(info.pool),
                                                 _hip.ptr(info.cnt[lo:hi]), b, K, mode_id, self.alpha, _hip.ptr(wsum),
                                                 _hip.ptr(pos), _hip.ptr(S), _hip.ptr(loss), _hip.ptr(ws),
                                                 _hip.stream_ptr()), "mkb_pool_step_bwd")
                total = loss if total is None else total + loss
        if m.name == "pRotatE" and self.world > 1:
            dist.all_reduce(m.modulus.grad, group=self.group)  # the modulus is replicated: its gradient sums over dims
        self._parts, self._info = parts, info
        return total.reshape(())

    @property
    def positive_score(self):
        return torch.cat([p[3] for p in self._parts]).view(-1, 1)

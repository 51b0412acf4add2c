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

__all__ = ["SparseGradExchange", "allreduce_touched_rows", "shard_rows"]


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
        if getattr(m.entity_embedding, "_mkb_lazy", None) is not None:
            # row-lazy Adam must step every row ANY rank touched: hand it the gathered id list
            m.entity_embedding._mkb_touched = self.last_union

    @property
    def last_union(self):
        return allreduce_touched_rows.last_union

    def loss(self, local_loss):
        t = local_loss.detach().reshape(1).clone()
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return t.reshape(())

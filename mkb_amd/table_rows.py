"""Row-sharded entity table across the GPUs of one node (BASELINE config 5: YAGO3-10, 123 k entities, RotatE-500; SURVEY
8(e) row 3).  The reference has no distributed code; this is the partitioning ``north_star`` names:

* rank g OWNS the entity rows e with ``e % world == g`` (local index ``e // world``): 1/world of the table, of its dense
  gradient and of the optimizer state.  The relation table (a few rows) is replicated;
* the global batch is cut by rows (rank g scores rows ``[g*B/world, (g+1)*B/world)``) against ONE candidate pool -- every
  rank replays the same MT19937 stream, so the negatives are bit-identical to a single-process run;
* per step, in this order
    1. pool rows  ``[P, De]``: every owner fills in the rows it holds, ONE all-reduce of the (disjoint) block makes it
       complete everywhere (an all-gather with uneven ownership; 2 MB at P = 512, De = 1000);
    2. positive rows: the heads / tails of this rank's triples are requested from their owners -- all-to-all of the id
       lists, all-to-all of the rows;
    3. local compute on the COMPACT table ``[pool rows | heads | tails]`` (``P + 2b`` rows) with the ids of the batch
       remapped into it: the fused HIP step (``mkb_pool_step``) runs unchanged, it never sees the global table;
    4. pool-row gradients + the relation gradient + the loss share: ONE all-reduce ("RCCL all-reduce of the sparse
       gradients"); each owner adds the rows it holds into its gradient shard;
    5. positive-row gradients travel back along the routes of 2 (all-to-all) and are scatter-added by their owners;
  then every rank steps dense Adam on its own shard: no optimizer communication.

Messages are a few MB at most and latency-bound; on xGMI's full mesh the direct all-to-all uses all 7 links at once.
Everything here is device-agnostic torch code: tests/test_parallel_gloo.py runs it on CPU (world 2 and 4, gloo) with the
oracle as the compute step and checks it against the single-process step.
"""
import torch
import torch.distributed as dist

__all__ = ["RowShardedTable", "TableRowShardedStep", "gather_table_rows", "shard_table_rows"]


def _world(group):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group):
    return dist.get_rank(group) if dist.is_initialized() else 0


class RowShardedTable:
    """``data`` = the rows this rank owns of a ``[n_rows, dim]`` table: global row e lives on rank ``e % world`` at local
    index ``e // world``.  ``grad`` is the matching shard of the dense gradient (allocated on first use)."""

    def __init__(self, n_rows, data, group=None):
        self.n_rows, self.group = int(n_rows), group
        self.rank, self.world = _rank(group), _world(group)
        self.data = data if isinstance(data, torch.nn.Parameter) else torch.nn.Parameter(data)
        want = (self.n_rows - self.rank + self.world - 1) // self.world
        if self.data.shape[0] != want:
            raise ValueError(f"rank {self.rank} owns {want} of {self.n_rows} rows, got a shard of {self.data.shape[0]}")

    @classmethod
    def from_full(cls, full, group=None, device=None):
        rank, world = _rank(group), _world(group)
        shard = full.detach()[rank::world].clone()
        return cls(full.shape[0], shard if device is None else shard.to(device), group)

    @property
    def dim(self):
        return self.data.shape[1]

    def _grad(self):
        if self.data.grad is None:
            self.data.grad = torch.zeros_like(self.data)
        return self.data.grad

    # ---- rows every rank needs (the candidate pool: the same ids, in the same order, on every rank)
    def gather_shared(self, ids):
        out = torch.zeros((ids.numel(), self.dim), dtype=self.data.dtype, device=self.data.device)
        mine = (ids % self.world) == self.rank
        out[mine] = self.data.detach()[torch.div(ids[mine], self.world, rounding_mode="floor")]
        if self.world > 1:
            dist.all_reduce(out, group=self.group)  # disjoint supports: the sum IS the gather, exactly
        return out

    def scatter_add_shared(self, ids, grad_rows):
        """``grad_rows`` already summed over the ranks: every owner adds the rows it holds (duplicates in ``ids`` add)."""
        mine = (ids % self.world) == self.rank
        self._grad().index_add_(0, torch.div(ids[mine], self.world, rounding_mode="floor"), grad_rows[mine])

    # ---- rows only this rank needs (the heads / tails of its own triples)
    def gather_private(self, ids):
        """-> (rows ``[len(ids), dim]``, route).  ``route`` brings gradients back with ``scatter_add_private``."""
        world, dev = self.world, ids.device
        owner = ids % world
        order = torch.argsort(owner, stable=True)           # requests grouped by owner, original order inside a group
        send_ids = torch.div(ids[order], world, rounding_mode="floor")
        send_counts = torch.bincount(owner, minlength=world)
        if world == 1:
            rows = self.data.detach()[send_ids]
            out = torch.empty_like(rows)
            out[order] = rows
            return out, (order, send_ids, None, None)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()  # (host sync: the split sizes of the next two collectives)
        want = torch.empty(sum(rc), dtype=torch.int64, device=dev)
        dist.all_to_all_single(want, send_ids, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        reply = self.data.detach()[want]                      # rows the others asked this owner for
        got = torch.empty((ids.numel(), self.dim), dtype=self.data.dtype, device=dev)
        dist.all_to_all_single(got, reply, output_split_sizes=sc, input_split_sizes=rc, group=self.group)
        out = torch.empty_like(got)
        out[order] = got
        return out, (order, want, sc, rc)

    def scatter_add_private(self, route, grad_rows):
        order, want, sc, rc = route
        grouped = grad_rows[order].contiguous()
        if sc is None:
            self._grad().index_add_(0, want, grouped)
            return
        back = torch.empty((want.numel(), self.dim), dtype=grad_rows.dtype, device=grad_rows.device)
        dist.all_to_all_single(back, grouped, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        self._grad().index_add_(0, want, back)


def shard_table_rows(model, group=None, device=None):
    """-> (entity ``RowShardedTable``, replicated relation ``Parameter``) from a full ``mkb_amd`` (or oracle-style) model
    that every rank built identically (same seed)."""
    table = RowShardedTable.from_full(model.entity_embedding, group, device)
    rel = model.relation_embedding.detach().clone()
    return table, torch.nn.Parameter(rel if device is None else rel.to(device))


def gather_table_rows(table):
    """Reassemble the full table from the shards (checkpointing, evaluation, tests)."""
    if table.world == 1:
        return table.data.detach().clone()
    per = (table.n_rows + table.world - 1) // table.world
    padded = torch.zeros((per, table.dim), dtype=table.data.dtype, device=table.data.device)
    padded[: table.data.shape[0]] = table.data.detach()
    parts = [torch.empty_like(padded) for _ in range(table.world)]
    dist.all_gather(parts, padded, group=table.group)
    full = torch.stack(parts, dim=1).reshape(per * table.world, table.dim)  # row e = parts[e % world][e // world]
    return full[: table.n_rows].contiguous()


class TableRowShardedStep:
    """``loss = step(sample, weight, negative_sample, mode)`` for this rank's rows of the global batch; fills
    ``table.data.grad`` (this rank's shard of the dense entity gradient) and ``relation.grad`` (replicated, already
    summed) and returns the GLOBAL loss.

    ``compute(ent, rel, sample, weight, pool_info, mode, weight_sum) -> (loss_share, g_ent, g_rel)`` runs the training
    step on the compact table; the default is the fused HIP step on a working model of ``model_cls``.  ``pool_info``
    carries ``pos [b, K]`` (slot -> compact row) and ``cnt [b, P]``; compact row p < P is pool position p."""

    def __init__(self, table, relation, alpha, model_cls=None, hidden_dim=None, gamma=None, group=None, compute=None,
                 modulus=None):
        self.table, self.relation, self.alpha, self.group = table, relation, float(alpha), group
        self.world = table.world
        self.compute = compute
        self._model_cls, self._hidden, self._gamma, self._modulus = model_cls, hidden_dim, gamma, modulus
        self._work = {}
        self._trains_modulus = getattr(model_cls, "__name__", "") == "pRotatE"

    # -- default compute: the fused pooled step (mkb_pool_step) on a working model that holds the compact table
    def _working_model(self, n_rows, device):
        m = self._work.get(n_rows)
        if m is None:
            ents = {i: i for i in range(n_rows)}
            rels = {i: i for i in range(self.relation.shape[0])}
            m = self._model_cls(hidden_dim=self._hidden, entities=ents, relations=rels, gamma=self._gamma).to(device)
            m.relation_embedding = self.relation  # the replicated table itself: its .grad is the relation gradient
            if self._modulus is not None:
                m.modulus = self._modulus
            elif self._trains_modulus:
                raise ValueError("pRotatE trains its modulus: pass the replicated `modulus` Parameter to the step")
            self._work[n_rows] = m
        return m

    def _fused(self, ent, rel, sample, weight, info, mode, weight_sum):
        from .fused import FusedTrainStep
        from .sampling.negative_sampling import PoolInfo

        m = self._working_model(ent.shape[0], ent.device)
        with torch.no_grad():
            m.entity_embedding.copy_(ent)
        if m.entity_embedding.grad is not None:
            m.entity_embedding.grad.zero_()
        rel_before = None if rel.grad is None else rel.grad.clone()
        neg = info.pos.long()  # slot -> compact row (only its pool description is used)
        P = info.cnt.shape[1]
        neg._mkb_pool = PoolInfo(torch.arange(P, device=ent.device), info.pos, info.cnt, info.size, info.mode_id, sample)
        step = self._work.setdefault(("step", ent.shape[0]), FusedTrainStep(m, self.alpha))
        loss = step(sample, weight, neg, mode, weight_sum=weight_sum)
        g_rel = rel.grad if rel_before is None else rel.grad - rel_before
        return loss, m.entity_embedding.grad, g_rel.clone()

    def __call__(self, sample, weight, negative_sample, mode):
        info = negative_sample._mkb_pool
        tb, dev = self.table, sample.device
        b, P = sample.shape[0], info.pool.numel()
        pool_rows = tb.gather_shared(info.pool)                                    # 1
        pos_rows, route = tb.gather_private(torch.cat([sample[:, 0], sample[:, 2]]))  # 2
        w_sum = weight.sum().reshape(1)
        if self.world > 1:
            dist.all_reduce(w_sum, group=self.group)
        ent = torch.cat([pool_rows, pos_rows])
        ar = torch.arange(b, device=dev)
        compact = torch.stack([P + ar, sample[:, 1], P + b + ar], dim=1).contiguous()
        rel = self.relation
        run = self.compute or self._fused
        if self.compute is None:
            if rel.grad is None:
                rel.grad = torch.zeros_like(rel)
        mod = self._modulus if (self.compute is None and self._trains_modulus) else None
        mod_before = None if mod is None or mod.grad is None else mod.grad.clone()
        loss, g_ent, g_rel = run(ent, rel, compact, weight, info, mode, w_sum)      # 3
        # 4: pool-row gradients + relation gradient (+ pRotatE's modulus gradient) + loss share in ONE all-reduce
        parts = [g_ent[:P].reshape(-1), g_rel.reshape(-1)]
        if mod is not None:
            g_mod = (mod.grad if mod_before is None else mod.grad - mod_before).reshape(-1).clone()
            parts.append(g_mod)
        buf = torch.cat(parts + [loss.reshape(1).to(g_ent.dtype)])
        if self.world > 1:
            dist.all_reduce(buf, group=self.group)
        if mod is not None:  # replicated scalar: replace this rank's share by the sum
            mod.grad.add_((buf[-2:-1] - g_mod).view_as(mod.grad))
        n_pool = P * tb.dim
        tb.scatter_add_shared(info.pool, buf[:n_pool].view(P, tb.dim))
        g_rel_sum = buf[n_pool: n_pool + rel.numel()].view_as(rel)
        if self.compute is None:   # the fused step already added this rank's share into rel.grad: replace it by the sum
            rel.grad.add_(g_rel_sum - g_rel)
        else:
            rel.grad = g_rel_sum.clone() if rel.grad is None else rel.grad + g_rel_sum
        tb.scatter_add_private(route, g_ent[P:])                                    # 5
        return buf[-1]

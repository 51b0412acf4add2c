"""Torch restatement of the three row-movement kernels of mkb_amd/csrc/rows.hip (``mkb_amd.table_rows.HipRowOps``), for the
CPU ``gloo`` tests of the row-sharded table's PROTOCOL (tests/test_parallel_gloo.py).  Test infrastructure: the product has
no CPU implementation and never imports this."""
import torch


class TorchRowOps:
    def route(self, ids, world, row0=0, sample_layout=False, merge=True):
        """Requests for the same row share the slot of the FIRST such request; the first requests are grouped by owner, request
        order kept inside a group; `send` lists their shard indices and is padded with -1 to the number of requests."""
        n = ids.shape[0]
        req = torch.cat([ids[:, 0], ids[:, 2]]) if sample_layout else ids.reshape(-1)
        m = req.numel()
        pos = torch.arange(m)
        if merge:
            uniq, inv = torch.unique(req, return_inverse=True)
            first = torch.full((uniq.numel(),), m, dtype=torch.int64).scatter_reduce(0, inv, pos, reduce="amin")[inv]
        else:
            first = pos
        reps = pos[first == pos]                               # the first request of every distinct row, in request order
        owner = req[reps] % world
        order = torch.argsort(owner, stable=True)              # grouped by owner, request order kept inside a group
        place = torch.empty(m, dtype=torch.int64)
        place[reps[order]] = torch.arange(reps.numel())
        slot = place[first]
        send = torch.full((m,), -1, dtype=torch.int64)
        send[: reps.numel()] = torch.div(req[reps[order]], world, rounding_mode="floor")
        counts = torch.bincount(owner, minlength=world)
        compact = None
        if sample_layout:
            compact = torch.stack([row0 + slot[:n], ids[:, 1], row0 + slot[n:]], dim=1).contiguous()
        return send, slot.to(torch.int32), counts, compact

    @staticmethod
    def _rows(seg):
        ids, rows, world, rank, local = seg
        if world <= 0:
            return ids.clamp(min=0), ids >= 0  # (-1: an entry behind a merged list / "not mine": skipped)
        return torch.div(ids, world, rounding_mode="floor"), (ids % world) == rank

    def gather(self, shard, segs, weight=None, weight_sum=None, zero=None, occ=None):
        for seg in segs:
            idx, mine = self._rows(seg)
            seg[1].zero_()
            seg[1][mine] = shard[idx[mine]]
            if seg[4] is not None:
                seg[4].copy_(torch.where(mine, idx, torch.full_like(idx, -1)))
        if weight_sum is not None:
            weight_sum.copy_(weight.sum().reshape(1))
        if zero is not None:
            zero.zero_()

    def scatter_add(self, grad, segs, dense_dst=None, dense_src=None, occ=None, copy_dst=None, copy_src=None):
        if copy_src is not None:
            copy_dst.copy_(copy_src)
        for seg in segs:
            idx, mine = self._rows(seg)
            grad.index_add_(0, idx[mine], seg[1][mine])
        if dense_src is not None:
            dense_dst.add_(dense_src)

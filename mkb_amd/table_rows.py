"""Row-sharded entity table across the GPUs of one node (BASELINE config 5: YAGO3-10, 123 k entities, RotatE-500; SURVEY
8(e) row 3).  The reference has no distributed code; this is the partitioning ``north_star`` names:

* rank g OWNS the entity rows e with ``e % world == g`` (shard index ``e // world``): 1/world of the table, of its dense
  gradient and of the optimizer state (``mkb_amd.optim.Adam(lazy_rows=True)`` steps the shard row-lazily, exactly like the
  single-GPU path).  The relation table (a few rows) is replicated;
* the global batch is cut by rows (every rank scores its own triples) against ONE candidate pool -- every rank replays the
  same MT19937 stream, so the negatives are bit-identical to a single-process run;
* per step
    0. (one batch AHEAD) ``mkb_rows_route`` merges the rank's 2b positive-row requests by row (a hub entity is requested once,
       however many of the batch's triples hold it) and groups them by owner; the per-owner counts are
       exchanged and read back on a side stream, the id lists follow -- so the all-to-alls of step t have host-known split
       sizes without the host ever waiting on the compute stream;
    1. the owners bring the rows about to be read up to date (``mkb_adam_rows_advance_sharded``) and read them
       (``mkb_rows_gather``: pool rows they hold -- zero rows otherwise -- straight into the compact table, requested rows
       into the all-to-all's send buffer; the same launch sums the batch's weights and clears the compact gradient);
    2. ONE all-reduce completes the pool block everywhere (disjoint supports: the sum IS the gather; the weight sum rides in
       a spare row), ONE all-to-all delivers the positive rows into the compact table ``[pool | spare | heads, tails]``;
    3. the fused HIP step (``mkb_pool_step``) runs UNCHANGED on the compact table with the triples re-addressed into it;
    4. ONE all-reduce sums pool-row gradients + relation gradient + loss share ("RCCL all-reduce of the sparse gradients":
       they live in one contiguous block of the compact gradient), ONE all-to-all returns the positive-row gradients;
    5. ``mkb_rows_scatter_add`` adds both into the owner's gradient shard (and the relation gradient into ``relation.grad``);
  the optimizer then steps each shard locally: no optimizer communication.

Messages are a few MB at most and latency-bound; on xGMI's full mesh the direct all-to-all uses all 7 links at once.

Backends.  The row movement goes through a small ``ops`` object: ``HipRowOps`` (the product: the three kernels of
``mkb_amd/csrc/rows.hip``) -- there is no CPU implementation in this package; tests that exercise the protocol on CPU
(``gloo``, oracle as the compute step) inject their own torch restatement (``tests/row_ops_torch.py``).
"""

import os

import torch
import torch.distributed as dist

from . import _hip, _links

__all__ = ["HipRowOps", "RowShardedTable", "TableRowShardedStep", "gather_table_rows", "shard_table_rows"]


def _world(group):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group):
    return dist.get_rank(group) if dist.is_initialized() else 0


def _collectives_run(world):
    """World 1 short-circuits every collective (the sum over one rank IS the operand) -- unless MKB_ROWS_FORCE_COLLECTIVES=1
    and a process group exists: then a 1-GPU box drives every call of the step (all_to_all_single with split sizes, the packed
    all-reduces, the side-stream read-back of the counts) through the real backend (``nccl`` = RCCL), which is how this path
    is exercised on hardware where only one GPU can be reached (tests/test_gpu_rccl_world1.py)."""
    return world > 1 or (os.environ.get("MKB_ROWS_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized())


def _lib_collectives_wanted():
    """MKB_ROWS_PY_COLLECTIVES=1 keeps the step's collectives in torch.distributed (the round-4 form: the A/B switch, and the
    only form on CPU / gloo)."""
    return os.environ.get("MKB_ROWS_PY_COLLECTIVES", "0") != "1"


class RowsComm:
    """The library's own RCCL communicators for the row-sharded step (``mkb_rows_comm_*``, mkb_amd/csrc/rows_comm.hip): the
    step's collectives are issued by ``libmkb_hip.so`` on the step's stream, the look-ahead planning on a side stream, and
    the all-to-alls' split sizes reach the host through a mailbox the device writes -- no ``torch.distributed`` call, no event
    wait and no ``.tolist()`` in the loop.  ``torch.distributed`` only carries the 256-byte id blob once, at set-up."""

    SLOTS = 4

    @classmethod
    def loopback(cls, hub, rank, world, device, max_requests):
        """A communicator over the library's IN-PROCESS transport (``mkb_rows_loop_hub_create``): every rank is a host thread of
        this process with its own streams on ``device``.  RCCL refuses two ranks on one device; this is how the step's glue is
        driven for world > 1 on a one-GPU box (tests/test_gpu_rows_loopback.py)."""
        import ctypes

        self = cls.__new__(cls)
        self.world, self.rank, self.device, self.max_requests = int(world), int(rank), device, int(max_requests)
        handle = ctypes.c_void_p()
        with _hip.on_device(device):
            _hip.check(_hip.lib().mkb_rows_comm_create_loopback(hub, self.rank, self.max_requests, ctypes.byref(handle)),
                       "mkb_rows_comm_create_loopback")
        self._handle = handle
        self.side = torch.cuda.Stream(device=device)
        self._n = 0
        self._I64 = ctypes.c_int64 * self.world
        return self

    def __init__(self, group, device, max_requests):
        import ctypes

        lib = _hip.lib()
        self.world, self.rank, self.device = _world(group), _rank(group), device
        cap = torch.tensor([int(max_requests)], dtype=torch.int64)
        blob = torch.zeros(256, dtype=torch.uint8)
        if self.rank == 0:
            _hip.check(lib.mkb_rows_comm_unique_id(blob.data_ptr()), "mkb_rows_comm_unique_id")
        if dist.is_initialized() and self.world >= 1:
            on_dev = dist.get_backend(group) == "nccl"
            src = dist.get_global_rank(group, 0) if group is not None else 0
            blob_t, cap_t = (blob.to(device), cap.to(device)) if on_dev else (blob, cap)
            dist.broadcast(blob_t, src=src, group=group)
            dist.all_reduce(cap_t, op=dist.ReduceOp.MAX, group=group)  # the id blocks have ONE capacity on every rank
            blob, cap = blob_t.cpu(), cap_t.cpu()
        self.max_requests = int(cap.item())
        handle = ctypes.c_void_p()
        with _hip.on_device(device):
            _hip.check(lib.mkb_rows_comm_create(blob.data_ptr(), self.rank, self.world, self.max_requests, ctypes.byref(handle)),
                       "mkb_rows_comm_create")
        self._handle = handle
        self.side = torch.cuda.Stream(device=device)
        self._n = 0
        self._I64 = ctypes.c_int64 * self.world

    def next_slot(self):
        self._n += 1
        return (self._n - 1) % self.SLOTS

    def plan(self, slot, sample, row0, bufs, bad):
        b = sample.shape[0]
        # MKB_ROWS_PLAN_ON_STEP_STREAM=1: the plan's id exchange is queued on the STEP's stream instead of the side stream, so the two
        # communicators never have a kernel in flight at the same time on a device.  The default (side stream) lets the plan of batch
        # t + 2 run beside step t; two RCCL communicators progressing concurrently on every device is something this code could only
        # exercise at world 1 and over the in-process transport (tests/test_gpu_rows_loopback.py) -- the switch is the fallback for a
        # node where ranks were seen to schedule the two in different orders and stall.
        side = _hip.stream_ptr(self.device) if os.environ.get("MKB_ROWS_PLAN_ON_STEP_STREAM", "0") == "1" else self.side.cuda_stream
        with _hip.on_device(self.device):
            _hip.check(_hip.lib().mkb_rows_comm_plan(self._handle, slot, _hip.ptr(sample), b, row0, _hip.ptr(bufs["send_ids"]),
                                                     _hip.ptr(bufs["slot"]), _hip.ptr(bufs["counts"]), _hip.ptr(bufs["compact"]),
                                                     _hip.ptr(bufs["want"]), bufs["want"].numel(), _hip.ptr(bad),
                                                     _hip.stream_ptr(self.device), side), "mkb_rows_comm_plan")

    def take(self, slot):
        sent, wanted = self._I64(), self._I64()
        with _hip.on_device(self.device):
            _hip.check(_hip.lib().mkb_rows_comm_take(self._handle, slot, sent, wanted, _hip.stream_ptr(self.device)), "mkb_rows_comm_take")
        return sent, wanted

    def exchange(self, reduce, send, send_rows, recv, recv_rows, D):
        with _hip.on_device(self.device):
            _hip.check(_hip.lib().mkb_rows_comm_exchange(self._handle, _hip.ptr(reduce), 0 if reduce is None else reduce.numel(),
                                                         _hip.ptr(send), send_rows, _hip.ptr(recv), recv_rows, D,
                                                         _hip.stream_ptr(self.device)), "mkb_rows_comm_exchange")

    def stats(self):
        """-> dict(plans, takes_that_waited, waited_with_idle_stream): a take waits when its plan has not executed yet; only
        those that found the step's stream EMPTY meanwhile left the device without work (the others: the host ran ahead)."""
        import ctypes

        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _hip.check(_hip.lib().mkb_rows_comm_stats(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "mkb_rows_comm_stats")
        return {"plans": a.value, "takes_that_waited": b.value, "waited_with_idle_stream": c.value}

    def close(self):
        if getattr(self, "_handle", None) is not None:
            torch.cuda.synchronize(self.device)
            _hip.lib().mkb_rows_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _LibRoute:
    """A plan made by ``RowsComm.plan``: the same fields the step reads from ``_Route`` (``send_ids``, ``slot``, ``compact``,
    ``listed`` = [lead | want], ``sc`` / ``rc``), backed by the ring buffers of its plan slot."""

    def __init__(self, comm, slot, bufs, lead, n):
        self.comm, self.plan_slot, self.bufs, self.lead, self.n = comm, slot, bufs, lead, n
        self.send_ids, self.slot, self.counts, self.compact = bufs["send_ids"], bufs["slot"], bufs["counts"], bufs["compact"]
        self.sc = self.rc = self.want = self.listed = None

    def resolve(self, lead=0):
        if self.want is not None:
            return
        assert lead == self.lead
        self.sc, self.rc = self.comm.take(self.plan_slot)
        self.n_sent, self.n_wanted = sum(self.sc), sum(self.rc)
        self.listed = self.bufs["listed"][: lead + self.n_wanted]
        self.want = self.listed[lead:]


class HipRowOps:
    """Row movement on the device (``mkb_rows_route`` / ``mkb_rows_gather`` / ``mkb_rows_scatter_add``).  A segment is
    ``(ids, rows, world, rank, local_ids_out)``: ``world == 0`` -> ``ids`` are shard indices; ``world > 0`` -> global
    entity ids, only this rank's entries are touched.

    Ids the reference's gather would answer with ``IndexError`` (mkb/models/base.py:193-207) never touch memory: the kernels
    skip them and raise a device flag that ``check()`` turns into that ``IndexError`` (one read-back, when the caller asks:
    ``TableRowShardedStep.check()``, like ``model.check_ids()`` on the single-GPU path)."""

    def __init__(self):
        self._bad = {}

    def _flag(self, dev):
        flag = self._bad.get(dev)
        if flag is None:
            flag = self._bad[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        return flag

    def check(self):
        for dev, flag in self._bad.items():
            bits = int(flag.item())
            if bits:
                flag.zero_()
                what = [w for b, w in ((1, "a negative entity id was routed"), (2, "a row index outside the table shard was listed")) if bits & b]
                raise IndexError("row-sharded table: " + " and ".join(what) + " (skipped on the device)")

    @staticmethod
    def _segs(segs):
        arr = (_hip.RowSeg * max(1, len(segs)))()
        for i, (ids, rows, world, rank, local) in enumerate(segs):
            _hip.require_device(ids, rows)
            arr[i] = _hip.RowSeg(ids.data_ptr(), ids.numel(), rows.data_ptr(), world, rank, None if local is None else local.data_ptr())
        return arr

    def route(self, ids, world, row0=0, sample_layout=False):
        """-> (send_ids, slot int32, counts int64 [world], compact [b, 3] or None)."""
        _hip.require_device(ids)
        ids = _hip.contiguous(ids, torch.int64)
        n = ids.shape[0]
        m = 2 * n if sample_layout else n
        dev = ids.device
        send = torch.empty(m, dtype=torch.int64, device=dev)
        slot = torch.empty(m, dtype=torch.int32, device=dev)
        counts = torch.empty(world, dtype=torch.int64, device=dev)
        compact = torch.empty((n, 3), dtype=torch.int64, device=dev) if sample_layout else None
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_rows_route(_hip.ptr(ids), n, 1 if sample_layout else 0, world, row0, _hip.ptr(send),
                                                 _hip.ptr(slot), _hip.ptr(counts), _hip.ptr(compact), _hip.ptr(self._flag(dev)),
                                                 _hip.stream_ptr()),
                       "mkb_rows_route")
        return send, slot, counts, compact

    def gather(self, shard, segs, weight=None, weight_sum=None, zero=None, occ=None):
        """``occ`` ([shard rows] int32, zeroed once): count how often each shard row is listed, for the matching
        ``scatter_add`` of the same segments (which resets the counts)."""
        _hip.require_device(shard)
        with _hip.on_device(shard.device):
            _hip.check(_hip.lib().mkb_rows_gather(
                _hip.ptr(shard), shard.shape[0], shard.shape[1], self._segs(segs), len(segs), _hip.ptr(weight),
                0 if weight is None else weight.numel(), _hip.ptr(weight_sum), _hip.ptr(zero),
                0 if zero is None else zero.numel() * zero.element_size(), _hip.ptr(occ), _hip.ptr(self._flag(shard.device)),
                _hip.stream_ptr()),
                "mkb_rows_gather")

    def scatter_add(self, grad, segs, dense_dst=None, dense_src=None, occ=None, copy_dst=None, copy_src=None):
        _hip.require_device(grad)
        with _hip.on_device(grad.device):
            _hip.check(_hip.lib().mkb_rows_scatter_add(
                _hip.ptr(grad), grad.shape[0], grad.shape[1], self._segs(segs), len(segs), _hip.ptr(dense_dst),
                _hip.ptr(dense_src), 0 if dense_src is None else dense_src.numel(), _hip.ptr(copy_dst), _hip.ptr(copy_src),
                0 if copy_src is None else copy_src.numel(), _hip.ptr(occ), _hip.ptr(self._flag(grad.device)),
                _hip.stream_ptr()),
                "mkb_rows_scatter_add")


class RowShardedTable:
    """``data`` = the rows this rank owns of a ``[n_rows, dim]`` table: global row e lives on rank ``e % world`` at shard
    index ``e // world``.  ``grad`` is the matching shard of the dense gradient (allocated on first use)."""

    def __init__(self, n_rows, data, group=None, ops=None, rank=None, world=None):
        self.n_rows, self.group = int(n_rows), group
        # (rank / world given explicitly: ranks that are not torch.distributed processes -- the in-process transport of the tests)
        self.rank, self.world = (_rank(group), _world(group)) if world is None else (int(rank), int(world))
        self.data = data if isinstance(data, torch.nn.Parameter) else torch.nn.Parameter(data)
        self.ops = HipRowOps() if ops is None else ops
        want = (self.n_rows - self.rank + self.world - 1) // self.world
        if self.data.shape[0] != want:
            raise ValueError(f"rank {self.rank} owns {want} of {self.n_rows} rows, got a shard of {self.data.shape[0]}")

    @classmethod
    def from_full(cls, full, group=None, device=None, ops=None, rank=None, world=None):
        if world is None:
            rank, world = _rank(group), _world(group)
        shard = full.detach()[rank::world].clone()
        return cls(full.shape[0], shard if device is None else shard.to(device), group, ops, rank=rank, world=world)

    @property
    def dim(self):
        return self.data.shape[1]

    def _grad(self):
        if self.data.grad is None:
            self.data.grad = torch.zeros_like(self.data)
        return self.data.grad

    # ---- rows every rank needs (the candidate pool: the same ids, in the same order, on every rank)
    def gather_shared(self, ids):
        out = torch.empty((ids.numel(), self.dim), dtype=self.data.dtype, device=self.data.device)
        self.ops.gather(self.data.detach(), [(ids, out, self.world, self.rank, None)])
        if _collectives_run(self.world):
            dist.all_reduce(out, group=self.group)  # disjoint supports: the sum IS the gather, exactly
        return out

    def scatter_add_shared(self, ids, grad_rows):
        """``grad_rows`` already summed over the ranks: every owner adds the rows it holds (duplicates in ``ids`` add)."""
        self.ops.scatter_add(self._grad(), [(ids, grad_rows.contiguous(), self.world, self.rank, None)])

    # ---- rows only this rank needs (the heads / tails of its own triples): the convenience form, which reads the split
    #      sizes back at once (TableRowShardedStep plans them one batch ahead instead)
    def gather_private(self, ids):
        """-> (rows ``[len(ids), dim]``, route).  ``route`` brings gradients back with ``scatter_add_private``."""
        send_ids, slot, counts, _ = self.ops.route(ids, self.world)
        route = _Route(self, ids.numel(), send_ids, slot, counts, None)
        route.exchange_counts()
        route.resolve()
        got = torch.empty((ids.numel(), self.dim), dtype=self.data.dtype, device=self.data.device)
        reply = torch.empty((route.want.numel(), self.dim), dtype=self.data.dtype, device=self.data.device)
        self.ops.gather(self.data.detach(), [(route.want, reply, 0, 0, None)])
        route.rows_to_requesters(reply, got)
        return got[slot.long()], route

    def scatter_add_private(self, route, grad_rows):
        grouped = torch.zeros_like(grad_rows)
        grouped.index_add_(0, route.slot.long(), grad_rows)  # (requests for the same row share a slot: their gradients add)
        back = torch.empty((route.want.numel(), self.dim), dtype=grad_rows.dtype, device=grad_rows.device)
        route.rows_to_owners(grouped, back)
        self.ops.scatter_add(self._grad(), [(route.want, back, 0, 0, None)])


class _Route:
    """Where the positive rows of one batch live.  ``send_ids`` (shard indices, grouped by owner) and ``slot`` come from
    ``ops.route``; ``exchange_counts`` tells every owner how many rows each rank will ask it for and starts the read-back of
    both count vectors on a side stream; ``resolve`` (at the step that uses the route) turns them into the host-known split
    sizes of the all-to-alls and sends the id lists: ``want`` = the shard indices the other ranks ask this owner for."""

    _side = {}
    host_waits = 0  # resolve() calls that found the read-back of their counts still in flight (the host then waits for it)

    def __init__(self, table, n, send_ids, slot, counts, compact):
        self.table, self.n = table, n
        self.send_ids, self.slot, self.counts, self.compact = send_ids, slot, counts, compact
        self.sc = self.rc = self.want = None
        self._host = self._event = self._ready = None

    _ring, _ring_at = {}, 0

    @classmethod
    def _pinned(cls, n):
        """A pinned int64 buffer for the counts' read-back, from a small ring (a fresh pinned allocation per step costs the
        host hundreds of microseconds and may serialise with the device: the read-back then is not there when the next step
        asks for it).  Four buffers: a route lives for one step and its buffer is consumed at the start of the next."""
        ring = cls._ring.get(n)
        if ring is None:
            ring = cls._ring[n] = [torch.empty(n, dtype=torch.int64, pin_memory=True) for _ in range(4)]
        cls._ring_at = (cls._ring_at + 1) % 4
        return ring[cls._ring_at]

    @classmethod
    def side_stream(cls, dev):
        side = cls._side.get(dev)
        if side is None:
            side = cls._side[dev] = torch.cuda.Stream(device=dev)
        return side

    def exchange_counts(self):
        tb = self.table
        if not _collectives_run(tb.world):
            return
        both = torch.empty(2 * tb.world, dtype=torch.int64, device=self.counts.device)
        both[: tb.world] = self.counts
        work = dist.all_to_all_single(both[tb.world:], both[: tb.world], group=tb.group, async_op=True)
        if both.is_cuda:  # read back beside the compute stream: the host waits for THIS copy only, never for the step's kernels
            side = self.side_stream(both.device)
            self._host = self._pinned(2 * tb.world)
            with torch.cuda.stream(side):
                if work is not None:
                    work.wait()
                self._host.copy_(both, non_blocking=True)
                self._event = torch.cuda.Event()
                self._event.record(side)
            both.record_stream(side)
        else:
            if work is not None:
                work.wait()
            self._host = both

    def resolve(self, lead=0):
        """``lead``: leave that many int64 slots in front of ``want`` inside one buffer (``self.listed``): the step keeps the
        shard indices of its pool rows there, so that [pool rows | requested rows] is ONE id list without a copy."""
        if self.want is not None:
            return
        tb = self.table
        if self._ready is not None:  # the route was made on the side stream: the step's stream takes over from here
            main = torch.cuda.current_stream(self.send_ids.device)
            main.wait_event(self._ready)
            # ... and the route's buffers, which the allocator handed out on the SIDE stream, are now read by the step's
            # kernels: without this the next plan() -- issued on the side stream while those kernels may still be in flight --
            # could be given the same memory the moment this route is dropped (seen as a rare wrong loss under RCCL)
            for t in (self.send_ids, self.slot, self.counts, self.compact):
                if t is not None:
                    t.record_stream(main)
            self._ready = None
        if not _collectives_run(tb.world):
            self.sc, self.rc = [self.n], [self.n]
            self.listed = torch.empty(lead + self.n, dtype=torch.int64, device=self.send_ids.device)
            self.want = self.listed[lead:]
            self.want.copy_(self.send_ids)
            return
        if self._event is not None:
            if not self._event.query():
                _Route.host_waits += 1
            self._event.synchronize()
        host = self._host.tolist()
        self.sc, self.rc = host[: tb.world], host[tb.world:]
        self.listed = torch.empty(lead + sum(self.rc), dtype=torch.int64, device=self.send_ids.device)
        self.want = self.listed[lead:]
        # (requests for the same row were merged by the route kernel: send_ids holds sum(sc) rows, then -1 padding)
        dist.all_to_all_single(self.want, self.send_ids[: sum(self.sc)], output_split_sizes=self.rc, input_split_sizes=self.sc,
                               group=tb.group)

    # rows [sum(rc), D] read by this owner -> the requesters' buffers [n, D] (grouped order), and the way back
    def rows_to_requesters(self, reply, got, async_op=False):
        if not _collectives_run(self.table.world):
            got.copy_(reply)
            return None
        return dist.all_to_all_single(got[: sum(self.sc)], reply, output_split_sizes=self.sc, input_split_sizes=self.rc,
                                      group=self.table.group, async_op=async_op)

    def rows_to_owners(self, grouped, back, async_op=False):
        if not _collectives_run(self.table.world):
            back.copy_(grouped)
            return None
        return dist.all_to_all_single(back, grouped[: sum(self.sc)], output_split_sizes=self.rc, input_split_sizes=self.sc,
                                      group=self.table.group, async_op=async_op)


def shard_table_rows(model, group=None, device=None, ops=None, rank=None, world=None):
    """-> (entity ``RowShardedTable``, replicated relation ``Parameter``) from a full ``mkb_amd`` (or oracle-style) model
    that every rank built identically (same seed)."""
    table = RowShardedTable.from_full(model.entity_embedding, group, device, ops, rank=rank, world=world)
    rel = model.relation_embedding.detach().clone()
    return table, torch.nn.Parameter(rel if device is None else rel.to(device))


def gather_table_rows(table):
    """Reassemble the full table from the shards (checkpointing, evaluation, tests)."""
    opt = _links.owner(table.data)
    if opt is not None:
        opt.flush(table.data)  # a row-lazy optimizer may hold steps that have not been applied yet
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
    """``loss = step(sample, weight, negative_sample, mode, next_sample=None)`` for this rank's rows of the global batch;
    fills ``table.data.grad`` (this rank's shard of the dense entity gradient) and ``relation.grad`` (replicated, already
    summed) and returns the GLOBAL loss.  ``next_sample``: the triples of the NEXT call (or call ``plan(next_sample)``
    yourself): their routing is prepared while this step runs, so that the next call starts without a host round trip.

    ``compute(ent, rel, sample, weight, pool_info, mode, weight_sum) -> (loss_share, g_ent, g_rel)`` runs the training step
    on the compact table; the default is the fused HIP step (``mkb_pool_step``) writing straight into the buffers the
    collectives use.  ``pool_info`` carries ``pos [b, K]`` (slot -> compact row) and ``cnt [b, P]``; compact row p < P is pool
    position p."""

    def __init__(self, table, relation, alpha, model_cls=None, hidden_dim=None, gamma=None, group=None, compute=None,
                 modulus=None, comm=None):
        self.table, self.relation, self.alpha, self.group = table, relation, float(alpha), group
        self.world, self.ops = table.world, table.ops
        self.compute = compute
        self._model_cls, self._hidden, self._gamma, self._modulus = model_cls, hidden_dim, gamma, modulus
        self._bufs, self._models, self._plans = {}, {}, []
        self._occ = None
        self._comm, self._plan_bufs = comm, {}  # (comm: a ready ``RowsComm``, e.g. ``RowsComm.loopback``; default: made at the first plan)
        self._trains_modulus = getattr(model_cls, "__name__", "") == "pRotatE"
        if compute is None and self._trains_modulus and modulus is None:
            raise ValueError("pRotatE trains its modulus: pass the replicated `modulus` Parameter to the step")

    # ------------------------------------------------------------------ layout of the compact table
    def _layout(self, P, b):
        """Compact table rows: [0, P) pool positions | [P, P + X) spare | [P + X, P + X + 2b) positive rows.  The spare rows
        of the TABLE carry the all-reduced weight sum (row P, element 0); the spare rows of its GRADIENT carry the relation
        gradient, the loss share and pRotatE's modulus gradient, so that ONE all-reduce of rows [0, P + X) moves them all."""
        D = self.table.dim
        extra = self.relation.numel() + 2
        X = max(1, -(-extra // D))
        return D, X, P + X, P + X + 2 * b

    def _buffers(self, P, b, dev):
        key = (P, b, dev)
        bufs = self._bufs.get(key)
        if bufs is None:
            D, X, row0, rows = self._layout(P, b)
            ent = torch.zeros((rows, D), dtype=torch.float32, device=dev)
            grad = torch.zeros((rows, D), dtype=torch.float32, device=dev)
            spare = grad[P: row0].view(-1)
            n_rel = self.relation.numel()
            bufs = self._bufs[key] = dict(ent=ent, grad=grad, wsum=ent[P, :1], g_rel=spare[:n_rel].view_as(self.relation),
                                          loss=spare[n_rel: n_rel + 1], g_mod=spare[n_rel + 1: n_rel + 2],
                                          pool_ids=torch.arange(P, device=dev))
        return bufs

    # ------------------------------------------------------------------ routing, one batch ahead
    @staticmethod
    def _batch_key(sample, P):
        # the batch as the CALLER holds it (before any .contiguous() copy): a view of the same storage is the same batch
        return (sample.data_ptr(), tuple(sample.shape), tuple(sample.stride()), sample._version, P)

    LOOKAHEAD_MAX = 3  # plans in flight besides the one being consumed (RowsComm.SLOTS - 1)

    def _lib_comm(self, sample):
        """The library-issued collectives (``RowsComm``) when they apply: collectives run at all, the batch is on a ROCm
        device, an RCCL runtime is bound, and MKB_ROWS_PY_COLLECTIVES is not set.  Created at the first plan (collectively)."""
        if self._comm is None:
            # (an `nccl` group = one rank per GPU, which is what RCCL itself asks for; gloo groups -- CPU tests, several test
            # ranks on one GPU -- keep the torch.distributed form)
            ok = (_collectives_run(self.world) and sample.is_cuda and _lib_collectives_wanted() and isinstance(self.ops, HipRowOps)
                  and dist.is_initialized() and dist.get_backend(self.group) == "nccl" and bool(_hip.lib().mkb_rows_comm_available()))
            self._comm = RowsComm(self.group, sample.device, 2 * sample.shape[0]) if ok else False
        return self._comm or None

    def _ring(self, comm, slot, b, P, dev):
        key = (slot, b, P)
        bufs = self._plan_bufs.get(key)
        if bufs is None:
            cap = P + 2 * b * self.world  # [lead: the pool rows' shard indices | want: at most every request of every rank]
            listed = torch.empty(cap, dtype=torch.int64, device=dev)
            bufs = self._plan_bufs[key] = dict(send_ids=torch.empty(2 * b, dtype=torch.int64, device=dev),
                                               slot=torch.empty(2 * b, dtype=torch.int32, device=dev),
                                               counts=torch.empty(self.world, dtype=torch.int64, device=dev),
                                               compact=torch.empty((b, 3), dtype=torch.int64, device=dev),
                                               listed=listed, want=listed[P:])
        return bufs

    def plan(self, sample, pool_size=None):
        """Prepare the routing of ``sample``'s positive rows; plans are consumed in the order they were made: the next
        ``step(sample, ...)`` without a plan of its own must be for the OLDEST planned batch (the same tensor or a view of the
        same storage, unmodified).  Needs the pool size of that step (``2 * sampler.size``) to address the compact table;
        defaults to the last step's.

        Every rank must plan (or not plan) alike: a plan issues a collective, so ranks that disagreed would hang.  That is why a
        pending plan is never dropped silently -- a step for another batch raises (``drop_plan()`` discards them, collectively).
        On a ROCm device the route kernel and the id exchange run on a side stream, beside the step's own launches."""
        P = self._last_P if pool_size is None else pool_size
        if len(self._plans) >= self.LOOKAHEAD_MAX + 1:
            raise RuntimeError(f"more than {self.LOOKAHEAD_MAX + 1} batches planned ahead")
        key = self._batch_key(sample, P)
        kept = sample
        sample = sample if sample.is_contiguous() else sample.contiguous()
        _, _, row0, _ = self._layout(P, sample.shape[0])
        comm = self._lib_comm(sample)
        if comm is not None:
            slot = comm.next_slot()
            bufs = self._ring(comm, slot, sample.shape[0], P, sample.device)
            comm.plan(slot, sample, row0, bufs, self.ops._flag(sample.device))
            route = _LibRoute(comm, slot, bufs, P, 2 * sample.shape[0])
            self._plans.append((key, route, kept, sample))
            return route
        side = ready = None
        if sample.is_cuda and _collectives_run(self.world):  # (world 1 without collectives: the side stream's events cost more than the 5 us route kernel they hide: 0.246 -> 0.283 ms/step measured)
            side = _Route.side_stream(sample.device)
            side.wait_stream(torch.cuda.current_stream(sample.device))  # (the batch may have been produced on the step's stream)
        if side is not None:
            with torch.cuda.stream(side):
                send_ids, slot, counts, compact = self.ops.route(sample, self.world, row0, sample_layout=True)
                route = _Route(self.table, 2 * sample.shape[0], send_ids, slot, counts, compact)
                route.exchange_counts()
                ready = torch.cuda.Event()
                ready.record(side)
            for t in (sample, send_ids, slot, counts, compact):
                t.record_stream(side)
            route._ready = ready
        else:
            send_ids, slot, counts, compact = self.ops.route(sample, self.world, row0, sample_layout=True)
            route = _Route(self.table, 2 * sample.shape[0], send_ids, slot, counts, compact)
            route.exchange_counts()
        self._plans.append((key, route, kept, sample))  # (keeps the tensors alive: the key stays unique)
        return route

    def drop_plan(self):
        """Discard every pending plan (every rank must do so alike)."""
        self._plans = []

    def _plan_ahead(self, upcoming, P):
        """``upcoming``: the next batch, or a list of the next batches in order; those not planned yet are planned now."""
        if upcoming is None:
            return
        batches = list(upcoming) if isinstance(upcoming, (list, tuple)) else [upcoming]
        for i, nxt in enumerate(batches[: self.LOOKAHEAD_MAX]):
            if i < len(self._plans):
                if self._plans[i][0] != self._batch_key(nxt, P):
                    raise RuntimeError("next_sample does not continue the batches already planned (plans are consumed in order)")
                continue
            self.plan(nxt, P)

    def _route_for(self, sample, P):
        if self._plans:
            plan = self._plans.pop(0)
            if plan[0] != self._batch_key(sample, P):
                self._plans.insert(0, plan)
                raise RuntimeError("the row-sharded step was handed another batch than the one planned for it (plan(next_sample) / "
                                   "next_sample= must name the very next batch, unmodified, with the same pool size, on every rank); "
                                   "call drop_plan() on every rank to discard a plan")
            route = plan[1]
        else:
            self.plan(sample, P)
            route = self._plans.pop(0)[1]
        route.resolve(lead=P)
        return route

    # ------------------------------------------------------------------ the compute step on the compact table
    def _working_model(self, bufs, dev):
        m = self._models.get(id(bufs))
        if m is None:
            rows = bufs["ent"].shape[0]
            ents, rels = {i: i for i in range(rows)}, {i: i for i in range(self.relation.shape[0])}
            m = self._model_cls(hidden_dim=self._hidden, entities=ents, relations=rels, gamma=self._gamma).to(dev)
            # its tables ARE the communication buffers / the replicated parameters: nothing is copied per step
            m.entity_embedding = torch.nn.Parameter(bufs["ent"], requires_grad=False)
            m.relation_embedding = self.relation
            if self._modulus is not None:
                m.modulus = self._modulus
            self._models[id(bufs)] = m
        return m

    def _fused(self, bufs, compact, weight, info, mode, b, P):
        from .fused import _workspace

        dev = weight.device
        m = self._working_model(bufs, dev)
        K = info.size
        pos = torch.empty((b, 1), dtype=torch.float32, device=dev)
        S = torch.empty((b, P), dtype=torch.float32, device=dev)
        ws = _workspace(m, b, K)
        # (the compact gradient was cleared by this step's gather launch: every row the step writes starts from zero)
        gr = _hip.Grads(bufs["grad"].data_ptr(), bufs["g_rel"].data_ptr(), bufs["g_mod"].data_ptr() if self._trains_modulus else None, 1)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_pool_step(m._tables(), gr, _hip.ptr(compact), _hip.ptr(weight), _hip.ptr(bufs["pool_ids"]),
                                                _hip.ptr(info.cnt), b, K, _hip.mode_id(mode), self.alpha, _hip.ptr(bufs["wsum"]),
                                                _hip.ptr(pos), _hip.ptr(S), _hip.ptr(bufs["loss"]), _hip.ptr(ws),
                                                _hip.stream_ptr()), "mkb_pool_step")
        self.positive_score, self._S = pos, S

    def sampled(self, sample, weight, sampler, mode, next_sample=None):
        """``step(sample, weight, sampler.generate(sample, mode), mode, next_sample)`` with the sampler folded into the shard's
        optimizer launch when the shard steps row-lazily (``mkb_adam_rows_advance_sharded_generate``: filter of this rank's rows
        + draw of the next pool + catch-up of the rows about to be read, one launch); identical negatives.  They stay
        available as ``self.negative_sample``."""
        opt = _links.owner(self.table.data)
        if opt is None or self.compute is not None or sampler.size > 512 or not sample.is_cuda:
            neg = sampler.generate(sample, mode)
            self.negative_sample = neg
            return self(sample, weight, neg, mode, next_sample=next_sample)
        return self(sample, weight, None, mode, next_sample=next_sample, _sampler=sampler)

    def __call__(self, sample, weight, negative_sample, mode, next_sample=None, _sampler=None):
        tb, ops, dev = self.table, self.ops, sample.device
        b = sample.shape[0]
        P = 2 * _sampler.size if _sampler is not None else negative_sample._mkb_pool.pool.numel()
        self._last_P = P
        route = self._route_for(sample, P)  # (keyed on the batch as the caller holds it, before any copy)
        sample = sample if sample.is_contiguous() else sample.contiguous()
        weight = weight if weight.is_contiguous() else weight.contiguous()
        D, X, row0, rows = self._layout(P, b)
        bufs = self._buffers(P, b, dev)
        ent, grad = bufs["ent"], bufs["grad"]
        self._plan_ahead(next_sample, P)  # their id exchange overlaps this step's kernels
        want = route.want
        lib = route.comm if isinstance(route, _LibRoute) else None
        R = want.numel()
        # 1. owners: rows about to be read become current (row-lazy Adam), then are read
        opt = _links.owner(tb.data)
        touched = route.listed  # [P + R] shard indices written this step: pool rows (-1: not mine; filled in by the gather) | want
        if self._occ is None or self._occ.device != dev:
            self._occ = torch.zeros(tb.data.shape[0], dtype=torch.int32, device=dev)
        if _sampler is not None:  # sampler + catch-up of [owned pool rows | requested rows] in one launch
            negative_sample = _sampler.generate_with_sharded_catch_up(sample, mode, opt, tb.data, self.world, tb.rank, want)
            self.negative_sample = negative_sample
            opt._state(tb.data)["caught_up"] = (touched, opt._state(tb.data)["n"])
        info = negative_sample._mkb_pool
        if opt is not None and _sampler is None:
            opt.catch_up_sharded(tb.data, info.pool, self.world, self.rank_of_table, want)
            opt._state(tb.data)["caught_up"] = (touched, opt._state(tb.data)["n"])
        # (one rank and no forced collectives: the all-to-alls are the identity -- the owner IS the user --, so the rows are
        # read straight into the compact table and their gradients taken straight from it: no stand-in copies)
        direct = not _collectives_run(self.world)
        reply = ent[row0: row0 + R] if direct else torch.empty((R, D), dtype=torch.float32, device=dev)
        ops.gather(tb.data.detach(), [(info.pool, ent[:P], self.world, tb.rank, touched[:P]), (want, reply, 0, 0, None)],
                   weight=weight, weight_sum=bufs["wsum"], zero=grad, occ=self._occ)
        # 2. positive rows to their users, pool block (+ weight sum) completed everywhere
        if lib is not None:  # ONE group on the step's stream: owners send what is wanted, users receive what they sent for
            lib.exchange(ent[: P + 1], reply, route.rc, ent[row0:], route.sc, D)
        else:
            w_rows = None if direct else route.rows_to_requesters(reply, ent[row0:], async_op=True)
            w_pool = dist.all_reduce(ent[: P + 1], group=self.group, async_op=True) if _collectives_run(self.world) else None
            for w in (w_rows, w_pool):
                if w is not None:
                    w.wait()
        # 3. the training step on the compact table
        rel = self.relation
        if rel.grad is None:
            rel.grad = torch.zeros_like(rel)
        if self.compute is None:
            self._fused(bufs, route.compact, weight, info, mode, b, P)
        else:
            loss, g_ent, g_rel = self.compute(ent, rel, route.compact, weight, info, mode, bufs["wsum"])
            grad.copy_(g_ent)
            grad[P: row0].zero_()
            bufs["g_rel"].copy_(g_rel)
            bufs["loss"].copy_(loss.reshape(1))
        # 4. pool-row gradients + relation gradient + loss share (+ modulus gradient): one all-reduce; positive-row
        #    gradients back to their owners
        back = grad[row0: row0 + R] if direct else torch.empty((R, D), dtype=torch.float32, device=dev)
        if lib is not None:
            lib.exchange(grad[:row0], grad[row0:], route.sc, back, route.rc, D)
        else:
            w_back = None if direct else route.rows_to_owners(grad[row0:], back, async_op=True)
            w_sum = dist.all_reduce(grad[:row0], group=self.group, async_op=True) if _collectives_run(self.world) else None
            for w in (w_back, w_sum):
                if w is not None:
                    w.wait()
        # 5. owners add what they hold; the relation gradient joins relation.grad in the same launch
        #    and the loss leaves the step buffers (the next step clears them) as a rider of that launch too
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ops.scatter_add(tb._grad(), [(info.pool, grad[:P], self.world, tb.rank, None), (want, back, 0, 0, None)],
                        dense_dst=rel.grad, dense_src=bufs["g_rel"], occ=self._occ, copy_dst=loss, copy_src=bufs["loss"])
        if self._trains_modulus and self.compute is None:
            mod = self._modulus
            if mod.grad is None:
                mod.grad = torch.zeros_like(mod)
            mod.grad.add_(bufs["g_mod"].view_as(mod.grad))
        if opt is not None:
            _links.mark_touched(tb.data, touched)
        return loss.reshape(())

    def check(self):
        """Raise the reference's ``IndexError`` for ids outside the table that reached the row kernels since the last call (they
        were skipped on the device, never dereferenced); one small read-back, when the caller asks."""
        if hasattr(self.ops, "check"):
            self.ops.check()

    @property
    def rank_of_table(self):
        return self.table.rank

    @property
    def negative_score(self):
        raise AttributeError("the row-sharded step keeps pool scores only (self._S [b, P]); gather them with the sampler's pos map")

"""``mkb_amd.optim.Adam`` -- dense Adam with the element-wise semantics of ``torch.optim.Adam`` (the optimizer
the reference's training loops construct, README.md:123-126; stepped at compose/pipeline.py:238-240), running as
one HBM-streaming HIP kernel per parameter (``mkb_adam_step``) with ``optimizer.zero_grad()`` fused into it.

Opt-in: ``torch.optim.Adam`` keeps working with ``mkb_amd`` models (gradients are ordinary dense ``.grad``
tensors).  Dense means dense: every row of the tables moves every step through its moments, exactly like the
reference -- rows with a zero gradient still decay ``exp_avg`` / ``exp_avg_sq`` and step along the stale
momentum.

``Adam(..., lazy_rows=True)`` keeps those semantics bit for bit but defers the zero-gradient steps of the rows a
step does not touch (``mkb_adam_rows_catchup`` / ``mkb_adam_rows_step``, see mkb_amd/csrc/adam.hip): the fused
training step tells the optimizer which entity rows it reads (candidate pool + heads + tails); those rows are
brought up to date before the forward pass and take the real step afterwards; everything else is replayed when
it is next needed or at ``flush()`` (``compose.Pipeline`` / ``evaluation`` / ``model(...)`` outside the fused
step / ``model.embeddings`` / ``model.save`` flush automatically; flush by hand before reading
``model.entity_embedding`` directly).

``defer_step=True`` (``compose.Pipeline`` turns it on for its fused loop) defers the REAL step of the touched rows as
well: after backward such a row is "current through t-1, gradient of step t in its ``.grad`` row", and the next
catch-up / flush that visits it replays step t WITH that gradient (then clears it) -- ``step()`` launches nothing, and
the second pass over p / m / v of every touched row (``mkb_adam_rows_step``) disappears.  Same arithmetic in the same
order: still bit-identical to dense Adam after ``flush()``.  The small dense parameters that rode the step launch (the
relation table) ride the next catch-up launch instead, so with ``defer_step`` they too are only current after
``flush()``.  Contract: clear gradients through ``optimizer.zero_grad()`` only (``model.zero_grad()`` /
``p.grad.zero_()`` would erase a step that has not been applied yet), and let every backward pass that writes the
table's gradient be preceded by ``catch_up`` of its rows (``FusedTrainStep`` / ``parallel.DimShardedStep`` do).
"""
import torch

from . import _hip, _links

__all__ = ["Adam"]


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, lazy_rows=False, draw_ahead=None, defer_step=None):
        """``draw_ahead``: a ``mkb_amd.sampling.NegativeSampling`` whose NEXT pool draw should ride the row catch-up
        launch of each step (one kernel launch and ~13 us of serial latency fewer per training step; the negatives are
        the same, bit for bit).  Only meaningful with ``lazy_rows=True`` and a sampler on the same device / stream."""
        self.params = [p for p in params]
        self.draw_ahead = draw_ahead
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}
        self.step_count = 0
        self.lazy_rows = lazy_rows
        self.defer_step = defer_step  # None: off until compose.Pipeline (or the caller) turns it on; False: never
        self._pending_dense = None    # (parameter, AdamDense, lr, gradient tensor kept alive) of a deferred step
        self._zeroed = False
        # the backward functions of model(...) add their rows straight into .grad of a row-lazily stepped table (no dense buffer
        # goes through autograd: _gradshare.direct); False: they hand autograd a dense gradient as for any other optimizer
        self.direct_grads = True
        if lazy_rows:
            for p in self.params:
                if p.dim() == 2 and p.shape[0] >= 4096:  # big tables only; small ones stay on the dense kernel
                    _links.attach(p, self)
                    if p.grad is None and p.is_cuda:
                        p.grad = torch.zeros_like(p)  # (where the row gradients land; all-zero outside pending rows from now on)

    # ------------------------------------------------------------------ state
    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            st = self.state[p] = {"m": torch.zeros_like(p), "v": torch.zeros_like(p), "n": 0}
            if _links.owner(p) is self:
                st["last"] = torch.zeros(p.shape[0], dtype=torch.int32, device=p.device)
                st["consts"] = torch.zeros((1 << 16, 2), dtype=torch.float32, device=p.device)
        return st

    def _consts(self, st, step):
        c = st["consts"]
        if step >= c.shape[0]:
            new = torch.zeros((max(2 * c.shape[0], step + 1), 2), dtype=torch.float32, device=c.device)
            new[: c.shape[0]] = c
            st["consts"] = c = new
        return c

    # ------------------------------------------------------------------ lazy rows
    def _take_dense(self):
        """The dense rider of the last deferred step (or None): it joins the launch that is about to be made."""
        pend, self._pending_dense = self._pending_dense, None
        return None if pend is None else pend[1]

    def _lr_of(self, st, step):
        return st.get("lrs", {}).get(step, self.lr)

    def catch_up(self, p, ids, upto=None):
        """Make the rows ``ids`` of ``p`` current through step ``upto`` (default: every step taken so far)."""
        st = self._state(p)
        upto = st["n"] if upto is None else upto
        if upto <= 0:
            return
        ids = _hip.contiguous(ids, torch.int64)
        st["caught_up"] = (ids, upto)
        lib, c = _hip.lib(), self._consts(st, upto)
        with _hip.on_device(p.device):
            if st.get("defer"):
                _hip.check(lib.mkb_adam_rows_advance(_hip.ptr(p.data), _hip.ptr(st["g"]), _hip.ptr(st["m"]), _hip.ptr(st["v"]),
                                                     _hip.ptr(st["last"]), _hip.ptr(c), p.shape[0], p.shape[1], _hip.ptr(ids),
                                                     ids.numel(), upto, self._lr_of(st, upto), self.betas[0], self.betas[1],
                                                     self.eps, self._take_dense(), self._sampler_handle(p.device),
                                                     _hip.stream_ptr()), "mkb_adam_rows_advance")
            else:
                _hip.check(lib.mkb_adam_rows_catchup(_hip.ptr(p.data), _hip.ptr(st["m"]), _hip.ptr(st["v"]),
                                                     _hip.ptr(st["last"]), _hip.ptr(c), p.shape[0], p.shape[1], _hip.ptr(ids),
                                                     ids.numel(), upto, self.betas[0], self.betas[1], self.eps,
                                                     self._sampler_handle(p.device), _hip.stream_ptr()),
                           "mkb_adam_rows_catchup")

    def catch_up_sharded(self, p, global_ids, world, rank, local_ids):
        """``catch_up`` for a ROW SHARD of a table (``mkb_amd.table_rows``): the rows to make current are the entries of
        ``global_ids`` this rank owns (``e % world == rank``, shard index ``e // world``) plus the shard indices
        ``local_ids``; one launch, no id list is materialised on the way."""
        st = self._state(p)
        upto = st["n"]
        if upto <= 0:
            return
        st["caught_up"] = (None, upto)
        lib, c = _hip.lib(), self._consts(st, upto)
        defer = bool(st.get("defer"))
        n_loc = 0 if local_ids is None else local_ids.numel()
        with _hip.on_device(p.device):
            _hip.check(lib.mkb_adam_rows_advance_sharded(
                _hip.ptr(p.data), _hip.ptr(st["g"]) if defer else None, _hip.ptr(st["m"]), _hip.ptr(st["v"]), _hip.ptr(st["last"]),
                _hip.ptr(c), p.shape[0], p.shape[1], _hip.ptr(global_ids), global_ids.numel(), world, rank,
                _hip.ptr(local_ids) if n_loc else None, n_loc, upto, self._lr_of(st, upto), self.betas[0], self.betas[1],
                self.eps, self._take_dense() if defer else None, self._sampler_handle(p.device), _hip.stream_ptr()),
                "mkb_adam_rows_advance_sharded")

    def catch_up_sharded_generate(self, p, world, rank, local_ids, sampler_handle, sample, B, mode_id, neg, pool, pos, cnt, touched):
        """``catch_up_sharded(p, the sampler's pool, world, rank, local_ids)`` fused with the sampler's filter of this rank's rows
        and the draw of the next pool: one launch (``mkb_adam_rows_advance_sharded_generate``)."""
        st = self._state(p)
        upto = st["n"]
        st["caught_up"] = (None, upto)
        lib, c = _hip.lib(), self._consts(st, max(upto, 1))
        defer = bool(st.get("defer"))
        n_loc = 0 if local_ids is None else local_ids.numel()
        with _hip.on_device(p.device):
            _hip.check(lib.mkb_adam_rows_advance_sharded_generate(
                _hip.ptr(p.data), _hip.ptr(st["g"]) if defer else None, _hip.ptr(st["m"]), _hip.ptr(st["v"]), _hip.ptr(st["last"]),
                _hip.ptr(c), p.shape[0], p.shape[1], world, rank, _hip.ptr(local_ids) if n_loc else None, n_loc, upto,
                self._lr_of(st, upto), self.betas[0], self.betas[1], self.eps, self._take_dense() if defer else None,
                sampler_handle, _hip.ptr(sample), B, mode_id, _hip.ptr(neg), _hip.ptr(pool), _hip.ptr(pos), _hip.ptr(cnt),
                _hip.ptr(touched), _hip.stream_ptr()), "mkb_adam_rows_advance_sharded_generate")

    def catch_up_generate(self, p, sampler_handle, sample, B, mode_id, neg, pool, pos, cnt, touched):
        """``catch_up(p, rows of this batch)`` fused with the sampler's filter + next-pool draw (one launch; see
        ``sampling.NegativeSampling.generate_with_catch_up``)."""
        st = self._state(p)
        upto = st["n"]
        lib, c = _hip.lib(), self._consts(st, max(upto, 1))
        tail = (sampler_handle, _hip.ptr(sample), B, mode_id, _hip.ptr(neg), _hip.ptr(pool), _hip.ptr(pos), _hip.ptr(cnt),
                _hip.ptr(touched), _hip.stream_ptr())
        with _hip.on_device(p.device):
            if st.get("defer"):
                _hip.check(lib.mkb_adam_rows_advance_generate(
                    _hip.ptr(p.data), _hip.ptr(st["g"]), _hip.ptr(st["m"]), _hip.ptr(st["v"]), _hip.ptr(st["last"]), _hip.ptr(c),
                    p.shape[0], p.shape[1], upto, self._lr_of(st, upto), self.betas[0], self.betas[1], self.eps,
                    self._take_dense(), *tail), "mkb_adam_rows_advance_generate")
            else:
                _hip.check(lib.mkb_adam_rows_catchup_generate(
                    _hip.ptr(p.data), _hip.ptr(st["m"]), _hip.ptr(st["v"]), _hip.ptr(st["last"]), _hip.ptr(c), p.shape[0],
                    p.shape[1], upto, self.betas[0], self.betas[1], self.eps, *tail), "mkb_adam_rows_catchup_generate")
        st["caught_up"] = (touched, upto)

    def flush(self, p=None):
        """Replay everything that is pending: afterwards the tables equal what dense Adam would hold."""
        lib = _hip.lib()
        for q in ([p] if p is not None else self.params):
            if _links.owner(q) is not self:
                continue
            st = self._state(q)
            if st["n"] <= 0 or st.get("flushed") == st["n"]:
                continue
            c = self._consts(st, st["n"])
            with _hip.on_device(q.device):
                if st.get("defer"):
                    _hip.check(lib.mkb_adam_rows_advance(_hip.ptr(q.data), _hip.ptr(st["g"]), _hip.ptr(st["m"]),
                                                         _hip.ptr(st["v"]), _hip.ptr(st["last"]), _hip.ptr(c), q.shape[0],
                                                         q.shape[1], None, 0, st["n"], self._lr_of(st, st["n"]), self.betas[0],
                                                         self.betas[1], self.eps, self._take_dense(), None, _hip.stream_ptr()),
                               "mkb_adam_rows_advance")
                else:
                    _hip.check(lib.mkb_adam_rows_catchup(_hip.ptr(q.data), _hip.ptr(st["m"]), _hip.ptr(st["v"]),
                                                         _hip.ptr(st["last"]), _hip.ptr(c), q.shape[0], q.shape[1], None, 0,
                                                         st["n"], self.betas[0], self.betas[1], self.eps, None,
                                                         _hip.stream_ptr()), "mkb_adam_rows_catchup")
            st["flushed"] = st["n"]
        pend, self._pending_dense = self._pending_dense, None
        if pend is not None:  # no launch above carried it (its table was already flushed): step it on its own
            q, d, lr, _ = pend
            with _hip.on_device(q.device):
                _hip.check(lib.mkb_adam_step(d.param, d.grad, d.exp_avg, d.exp_avg_sq, d.n, d.step, lr, self.betas[0],
                                             self.betas[1], self.eps, 1, _hip.stream_ptr()), "mkb_adam_step")

    def stop_deferring(self):
        """Apply whatever ``defer_step`` left pending and switch it off for good (callers whose gradient rows are not
        preceded by a catch-up, e.g. ``parallel.SparseGradExchange``)."""
        self.flush()
        self.defer_step = False
        for st in self.state.values():
            st.pop("defer", None)

    # ------------------------------------------------------------------ torch.optim-like API
    def _rider(self):
        """The dense tensor that steps inside the row-lazy launch (one fewer kernel per step): the first dense,
        16-byte aligned float32 parameter with a gradient, when some table steps row-lazily this time."""
        lazy = [p for p in self.params if p.grad is not None and _links.owner(p) is self and _links.touched(p) is not None
                and not _links.autograd_wrote(p)]  # (autograd wrote as well: that table takes the dense route this time)
        if len(lazy) != 1:
            return None, None
        for q in self.params:
            if (q.grad is not None and _links.owner(q) is not self and q.is_cuda and q.device == lazy[0].device
                    and q.dtype == torch.float32 and q.is_contiguous() and q.grad.is_contiguous() and q.numel() >= 4
                    and "last" not in self._state(q)):
                st = self._state(q)
                ptrs = (q.data_ptr(), q.grad.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr())
                if all(x % 16 == 0 for x in ptrs):
                    return q, _hip.AdamDense(*ptrs, q.numel(), st["n"] + 1)
        return None, None

    def _sampler_handle(self, device):
        s = self.draw_ahead
        if s is None or getattr(s, "_handle", None) is None or s._device != device:
            return None  # (the sampler creates its device state on its first generate())
        return s._handle

    def step(self):
        self.step_count += 1
        lib = _hip.lib()
        rider_p, rider = self._rider()
        dense_todo = []
        for p in self.params:
            if p.grad is None:
                continue
            if p is rider_p:  # stepped by the row-lazy launch below / above
                self._state(p)["n"] += 1
                continue
            _hip.require_device(p)
            st = self._state(p)
            g = p.grad
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            wrote = _links.autograd_wrote(p)
            current = _links.all_marks_current(p)  # (every row was made current by the forward pass that read it)
            touched = _links.take_touched(p) if _links.owner(p) is self else None
            if wrote:  # autograd accumulated into .grad in this step as well: which rows is unknown -> the dense route below
                touched = None
            with _hip.on_device(p.device):
                if touched is not None:
                    done = st.get("caught_up")
                    caught = done is not None and done[1] == st["n"]  # a catch-up at this step count preceded the gradient
                    if self.defer_step and st["n"] >= 1 and caught:
                        # deferred: the rows were current through n before their gradient was written; the next catch-up /
                        # flush that visits a row replays this step with its gradient row (nothing is launched here)
                        st["n"] += 1
                        st["defer"], st["g"] = True, g
                        lrs = st.setdefault("lrs", {})
                        lrs[st["n"]] = self.lr
                        lrs.pop(st["n"] - 4, None)
                        if rider_p is not None:
                            self._pending_dense = (rider_p, rider, self.lr, rider_p.grad)
                        continue
                    if st.get("defer") and not caught:
                        raise RuntimeError("mkb_amd.optim.Adam(defer_step=True): gradient rows were written without a catch-up of "
                                           "those rows in this step, so they cannot be told from a step that is still pending; "
                                           "call optimizer.stop_deferring() before training this way")
                    st["n"] += 1
                    ids = _hip.contiguous(touched, torch.int64)
                    c = self._consts(st, st["n"])
                    if not current and (done is None or done[0] is not touched or done[1] != st["n"] - 1):
                        self.catch_up(p, ids, upto=st["n"] - 1)  # e.g. rows only OTHER data-parallel ranks touched
                    _hip.check(lib.mkb_adam_rows_step(_hip.ptr(p.data), _hip.ptr(g), _hip.ptr(st["m"]), _hip.ptr(st["v"]),
                                                      _hip.ptr(st["last"]), _hip.ptr(c), p.shape[0], p.shape[1],
                                                      _hip.ptr(ids), ids.numel(), st["n"], self.lr, self.betas[0],
                                                      self.betas[1], self.eps, rider, _hip.stream_ptr()), "mkb_adam_rows_step")
                else:
                    if "last" in st:  # a step whose touched rows are unknown: fall back to dense for good
                        self.flush(p)
                        st.pop("last"), st.pop("consts")
                        _links.detach(p)
                    st["n"] += 1
                    dense_todo.append((p, g, st))
        # the dense tensors of a device step in ONE launch (mkb_adam_step_multi: the same arithmetic per element; small models
        # are bound by launches -- Umls TransE-64 spends 9 launches of ~5 us per step)
        while dense_todo:
            dev = dense_todo[0][0].device
            group = [t for t in dense_todo if t[0].device == dev][:8]
            dense_todo = [t for t in dense_todo if not any(t is u for u in group)]
            aligned = all(x.data_ptr() % 16 == 0 for p, g, st in group for x in (p.data, g, st["m"], st["v"]))
            sampler = self._sampler_handle(dev)  # (its next pool is drawn by one more workgroup of this launch)
            with _hip.on_device(dev):
                if (len(group) > 1 or sampler is not None) and aligned:
                    arr = (_hip.AdamDense * len(group))(*[_hip.AdamDense(p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(),
                                                                          p.numel(), st["n"]) for p, g, st in group])
                    _hip.check(lib.mkb_adam_step_multi(arr, len(group), self.lr, self.betas[0], self.betas[1], self.eps, 1,
                                                       sampler, _hip.stream_ptr()), "mkb_adam_step_multi")
                else:
                    for p, g, st in group:
                        _hip.check(lib.mkb_adam_step(_hip.ptr(p.data), _hip.ptr(g), _hip.ptr(st["m"]), _hip.ptr(st["v"]),
                                                     p.numel(), st["n"], self.lr, self.betas[0], self.betas[1], self.eps, 1,
                                                     _hip.stream_ptr()), "mkb_adam_step")
        self._zeroed = True

    def zero_grad(self, set_to_none=False):
        """Gradients were already cleared inside ``step`` (fused); they stay allocated so the next backward
        accumulates in place."""
        if self._zeroed and not set_to_none:
            self._zeroed = False
            return
        if any(st.get("defer") for st in self.state.values()):
            self.flush()  # a deferred step lives in the gradient rows: apply it before they are really cleared
        for p in self.params:
            _links.clear_autograd_wrote(p)
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
        self._zeroed = False

    # ------------------------------------------------------------------ checkpointing (torch.optim-style)
    def state_dict(self):
        """Like ``torch.optim.Adam.state_dict()``: per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` (clones) keyed by
        the parameter's position, plus the hyper-parameters.  Pending row-lazy steps are flushed first, so the tables and
        the moments the caller saves next to each other are the dense-Adam state of this step."""
        self.flush()
        state = {}
        for i, p in enumerate(self.params):
            st = self.state.get(p)
            if st is not None:
                state[i] = {"step": st["n"], "exp_avg": st["m"].clone(), "exp_avg_sq": st["v"].clone()}
        return {"state": state, "param_groups": [{"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                                                  "params": list(range(len(self.params)))}],
                "step_count": self.step_count}

    def load_state_dict(self, sd):
        """Restore ``state_dict()`` onto an optimizer built over the same parameters (their tables must be restored by the
        caller, e.g. ``model.load_state_dict``): every row is then current through the saved step."""
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self.step_count = sd.get("step_count", 0)
        for i, p in enumerate(self.params):
            saved = sd["state"].get(i, sd["state"].get(str(i)))
            if saved is None:
                self.state.pop(p, None)
                continue
            st = self._state(p)
            st["m"].copy_(saved["exp_avg"])
            st["v"].copy_(saved["exp_avg_sq"])
            st["n"] = int(saved["step"])
            st.pop("caught_up", None)
            st["flushed"] = st["n"]
            if st.pop("defer", None):  # nothing is pending in a restored state
                st["g"].zero_()
            if "last" in st:  # row-lazy: nothing is pending, the replay constants of earlier steps are not needed again
                st["last"].fill_(st["n"])
        self._pending_dense = None
        self._zeroed = False

"""``Pipeline`` -- the reference's training loop (mkb/compose/pipeline.py:183-327) with the same constructor,
``learn`` signature, call order, evaluation cadence, early-stopping rule and printed output.

Per batch the reference runs (pipeline.py:206-244):
    sample -> model(sample) -> sampling.generate -> model(sample, neg, mode) -> loss -> backward
    -> optimizer.step -> optimizer.zero_grad -> rolling mean of loss.item()
When model, sampler and loss are the ``mkb_amd`` ones, the forward/loss/backward part is ONE call into
``mkb_pool_step`` (``FusedTrainStep``) that writes the dense ``.grad`` buffers directly; anything else (a custom
loss, a foreign sampler, classification mode) takes the reference's explicit sequence through autograd, which
still dispatches to the HIP kernels.  The optimizer is the user's (``torch.optim.Adam`` in the reference's
README, or ``mkb_amd.optim.Adam``).
"""
import torch

from ..datasets.device import DeviceBatches
from ..fused import FusedTrainStep, pooled_supported
from ..losses import Adversarial
from ..models.base import BaseModel
from ..sampling import NegativeSampling
from ..utils import Bar, RollingMean

__all__ = ["Pipeline"]


_WATCHED = ("HITS@3", "HITS@1")  # a round only counts as "no improvement" when both fall (pipeline.py:272-299)


class _Patience:
    """Early-stopping bookkeeping for the watched split.  ``anchor`` is the score dict of the last round that was NOT
    worse (the reference re-anchors on every such round, it does not keep the maximum); ``stale`` counts the rounds in a
    row that fell below the anchor on every watched metric."""

    def __init__(self, limit):
        self.limit, self.anchor, self.stale = limit, None, 0

    def observe(self, scores):
        fell = self.anchor is not None and all(self.anchor[m] > scores[m] for m in _WATCHED)
        if fell:
            self.stale += 1
        else:
            self.stale, self.anchor = 0, scores

    @property
    def exhausted(self):
        return self.stale == self.limit  # (limit 0 stops at the first evaluation, as the reference's == test does)


class _HostStager:
    """Host batch -> device without stopping the host: the batch is copied into one of ``depth`` page-locked buffers and sent
    with a non-blocking copy on the current stream; an event per buffer says when it may be overwritten.  ``tensor.to(device)``
    of a pageable tensor waits until the copy has RUN, i.e. until the device has finished everything queued before it -- the
    host then prepares every step's launches with the device idle (the drop-in ``datasets`` loaders: 0.279 ms / step against
    0.217 with batches resident on the device)."""

    def __init__(self, device, depth=4):
        self.device, self.depth, self.k, self.slots = device, depth, 0, {}
        self.on = torch.device(device).type == "cuda"
        self.stream = torch.cuda.current_stream(torch.device(device)) if self.on else None

    def __call__(self, *tensors):
        """The tensors of ONE batch -> their device copies (one buffer set and one event per batch)."""
        if not self.on or any((not torch.is_tensor(t)) or t.is_cuda for t in tensors):
            return tuple(t.to(self.device) for t in tensors)
        key = (self.k % self.depth,) + tuple((t.dtype, tuple(t.shape)) for t in tensors)
        self.k += 1
        slot = self.slots.get(key)
        if slot is None:
            slot = self.slots[key] = [[torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors], torch.cuda.Event(), False]
        bufs, ev, used = slot
        if used:
            ev.synchronize()  # (the copies that last read these buffers have run)
        out = []
        for buf, t in zip(bufs, tensors):
            buf.copy_(t)
            out.append(buf.to(self.device, non_blocking=True))
        ev.record(self.stream)
        slot[2] = True
        return tuple(out)


class Pipeline:
    def __init__(self, epochs, eval_every=2000, early_stopping_rounds=3, device="cpu"):
        self.epochs, self.eval_every, self.early_stopping_rounds = epochs, eval_every, early_stopping_rounds
        self.device, self.metric_loss = device, RollingMean(1000)
        self.valid_scores, self.test_scores = {}, {}
        self.fuse = True  # set False to force the unfused autograd sequence
        # Opt-in: iterate ``datasets.DeviceBatches`` (training triples + weights resident in HBM, batches index-selected on
        # the device) instead of the dataset's two host DataLoaders.  Same batch format, alternation and coverage; the
        # shuffle ORDER is a seeded device randperm, not the reference's CPU RandomSampler order, hence not the default.
        self.device_batches = False

    # ------------------------------------------------------------------ one epoch of steps (pipeline.py:206-244)
    def _fused_step_for(self, model, dataset, sampling, optimizer, loss):
        """``FusedTrainStep`` when model, sampler and loss are the mkb_amd ones on a ROCm device, else None."""
        if not (self.fuse and isinstance(model, BaseModel) and isinstance(sampling, NegativeSampling)
                and type(loss) is Adversarial and model.entity_embedding.is_cuda
                and pooled_supported(model, dataset.batch_size, sampling.size)):
            return None
        self._borrowed = []  # optimizer settings this loop switches on and learn() switches back off before it returns
        if getattr(optimizer, "lazy_rows", False) and getattr(optimizer, "draw_ahead", "x") is None:
            optimizer.draw_ahead = sampling  # mkb_amd.optim.Adam: the next pool's draw rides the catch-up launch
            self._borrowed.append("draw_ahead")
        if getattr(optimizer, "lazy_rows", False) and getattr(optimizer, "defer_step", "x") is None:
            optimizer.defer_step = True  # this loop clears gradients through optimizer.zero_grad() only: the real step of
            #                              the touched rows may wait for the next catch-up launch (mkb_amd/optim.py)
            self._borrowed.append("defer_step")
        return FusedTrainStep(model, loss.alpha)

    def _give_back(self, optimizer):
        """Deferral is scoped to learn(): a pending step lives in the table's gradient rows, and the user's own code after
        learn() may clear gradients any way it likes (model.zero_grad(), p.grad = None) -- so nothing may be left pending and
        the optimizer goes back to torch.optim semantics (step() applies the step)."""
        borrowed, self._borrowed = getattr(self, "_borrowed", []), []
        if "defer_step" in borrowed:
            optimizer.stop_deferring()   # applies what is pending, then step() launches again
            optimizer.defer_step = None  # (a later learn() may borrow it again)
        if "draw_ahead" in borrowed:
            optimizer.draw_ahead = None

    def _run_epoch(self, epoch, fused, model, dataset, sampling, optimizer, loss):
        pending = []  # fused path: losses stay on the device until the bar refreshes (one D2H copy per 10 steps)

        def drain():
            if pending:
                for v in torch.stack(pending).tolist():
                    self.metric_loss.update(v)
                pending.clear()

        # the fused step runs where the model lives (a script that keeps the reference's default device="cpu" but moved the
        # model to the GPU still works); the explicit sequence follows the reference and uses self.device
        device = model.entity_embedding.device if fused is not None else self.device
        bar = Bar(dataset=dataset, update_every=10)
        to_device = _HostStager(device) if fused is not None else (lambda *ts: tuple(t.to(device) for t in ts))
        for data in bar:
            mode = data["mode"]
            if mode == "classification":
                raise NotImplementedError("classification mode (ConvE / BCE) is outside the mkb_amd hot path")
            sample, weight = to_device(data["sample"], data["weight"])
            if fused is not None:
                # generate + fused step; with a row-lazy mkb_amd.optim.Adam the sampler rides the catch-up launch
                error = fused.sampled(sample, weight, sampling, mode)
            else:  # the reference's explicit sequence (pipeline.py:211-236), every call on the HIP kernels
                positive_score = model(sample)
                negatives = sampling.generate(sample=sample, mode=mode).to(self.device)
                error = loss(positive_score, model(sample=sample, negative_sample=negatives, mode=mode), weight)
                error.backward()
            optimizer.step()
            optimizer.zero_grad()
            if fused is not None:
                # the reference syncs on error.item() every step (pipeline.py:242); the rolling mean only has to be
                # current when it is shown, so the same values are fed to it in the same order, in batches
                pending.append(error.detach())
                if bar.due() or len(pending) >= 64:
                    drain()
            else:
                self.metric_loss.update(float(error.item()))
            bar.set_description("Epoch: %d, loss: %4f" % (epoch, self.metric_loss.get()))
        drain()
        if hasattr(sampling, "check"):
            sampling.check()  # KeyError / empty-filter errors of this epoch's batches (lazy: no per-batch sync)
        if hasattr(model, "check_ids"):
            model.check_ids()  # IndexError for ids outside the tables (flagged on the device by the scoring calls)

    # ------------------------------------------------------------------ evaluation rounds (pipeline.py:246-321)
    def _score_splits(self, evaluation, model, dataset):
        """Entity + relation ranking metrics of every split the dataset holds, stored and printed."""
        for attr, title, triples in (("valid_scores", "Validation:", dataset.valid), ("test_scores", "Test:", dataset.test)):
            if triples:
                scores = evaluation.eval(model=model, dataset=triples)
                scores.update(evaluation.eval_relations(model=model, dataset=triples))
                setattr(self, attr, scores)
                self.print_metrics(description=title, metrics=scores)

    def learn(self, model, dataset, sampling, optimizer, loss, evaluation=None):
        if self.device_batches and not isinstance(dataset, DeviceBatches):
            dataset = DeviceBatches(dataset, device=self.device, seed=getattr(dataset, "seed", None) or 42)
        fused = self._fused_step_for(model, dataset, sampling, optimizer, loss)
        try:
            self._learn(fused, model, dataset, sampling, optimizer, loss, evaluation)
            if hasattr(model, "sync_parameters"):
                model.sync_parameters()  # a row-lazy / deferring optimizer leaves nothing pending behind learn()
        finally:
            # also on an exception / KeyboardInterrupt inside the loop: a deferred step still pending in the table's gradient
            # rows is applied and the optimizer goes back to torch.optim semantics before the caller's code sees it
            self._give_back(optimizer)
        return self

    def _learn(self, fused, model, dataset, sampling, optimizer, loss, evaluation):
        patience = _Patience(self.early_stopping_rounds)
        for epoch in range(self.epochs):
            self._run_epoch(epoch, fused, model, dataset, sampling, optimizer, loss)
            if evaluation is None or (epoch + 1) % self.eval_every != 0:
                continue
            print("\n Epoch: %d." % epoch)
            self._score_splits(evaluation, model, dataset)
            patience.observe(self.test_scores if dataset.test else self.valid_scores)  # the test split leads when present
            if patience.exhausted:
                print("\n Early stopping at epoch %d." % epoch)
                for title, scores in (("Validation:", self.valid_scores), ("Test:", self.test_scores)):
                    self.print_metrics(description=title, metrics=scores)
                break
        else:
            print("\n Epoch: %d. \n" % epoch)
            self._score_splits(evaluation, model, dataset)  # (the reference, too, needs an evaluation object here)

    @classmethod
    def print_metrics(cls, description, metrics):
        print("\n".join([f"\t {description}"] + [f"\t\t {name}: {value}" for name, value in metrics.items()]))

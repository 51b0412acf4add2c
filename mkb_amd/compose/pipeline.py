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
import collections

import torch

from ..fused import FusedTrainStep, pooled_supported
from ..losses import Adversarial
from ..models.base import BaseModel
from ..sampling import NegativeSampling
from ..utils import Bar, RollingMean

__all__ = ["Pipeline"]


class Pipeline:
    def __init__(self, epochs, eval_every=2000, early_stopping_rounds=3, device="cpu"):
        self.epochs = epochs
        self.eval_every = eval_every
        self.early_stopping_rounds = early_stopping_rounds
        self.device = device
        self.metric_loss = RollingMean(1000)
        self.round_without_improvement_valid = 0
        self.round_without_improvement_test = 0
        self.history_valid = collections.defaultdict(float)
        self.history_test = collections.defaultdict(float)
        self.valid_scores = {}
        self.test_scores = {}
        self.fuse = True  # set False to force the unfused autograd sequence

    def learn(self, model, dataset, sampling, optimizer, loss, evaluation=None):
        fused = None
        if (self.fuse and isinstance(model, BaseModel) and isinstance(sampling, NegativeSampling)
                and type(loss) is Adversarial and model.entity_embedding.is_cuda
                and pooled_supported(model, dataset.batch_size, sampling.size)):
            fused = FusedTrainStep(model, loss.alpha)
            if getattr(optimizer, "lazy_rows", False) and getattr(optimizer, "draw_ahead", "x") is None:
                optimizer.draw_ahead = sampling  # mkb_amd.optim.Adam: the next pool's draw rides the catch-up launch

        pending = []  # fused path: losses stay on the device until the bar refreshes (one D2H copy per 10 steps)

        def flush():
            if pending:
                for v in torch.stack(pending).tolist():
                    self.metric_loss.update(v)
                pending.clear()

        for epoch in range(self.epochs):
            bar = Bar(dataset=dataset, update_every=10)
            for data in bar:
                sample = data["sample"].to(self.device)
                mode = data["mode"]
                if mode == "classification":
                    raise NotImplementedError("classification mode (ConvE / BCE) is outside the mkb_amd hot path")
                weight = data["weight"].to(self.device)
                if fused is not None:
                    # generate + fused step; with a row-lazy mkb_amd.optim.Adam the sampler rides the catch-up launch
                    error = fused.sampled(sample, weight, sampling, mode)
                    negative_sample = fused.negative_sample
                else:
                    score = model(sample)
                    negative_sample = sampling.generate(sample=sample, mode=mode)
                    negative_sample = negative_sample.to(self.device)
                    negative_score = model(sample=sample, negative_sample=negative_sample, mode=mode)
                    error = loss(score, negative_score, weight)
                    error.backward()
                _ = optimizer.step()
                optimizer.zero_grad()
                if fused is not None:
                    # the reference syncs on error.item() every step (pipeline.py:242); the rolling mean only has to be
                    # current when it is shown, so the same values are fed to it in the same order, in batches
                    pending.append(error.detach())
                    if bar.due() or len(pending) >= 64:
                        flush()
                else:
                    self.metric_loss.update(error.item())
                bar.set_description(f"Epoch: {epoch}, loss: {self.metric_loss.get():4f}")
            flush()

            if hasattr(sampling, "check"):
                sampling.check()  # KeyError / empty-filter errors of this epoch's batches (lazy: no per-batch sync)

            if evaluation is not None:
                if (epoch + 1) % self.eval_every == 0:
                    print(f"\n Epoch: {epoch}.")
                    if dataset.valid:
                        self.valid_scores = evaluation.eval(model=model, dataset=dataset.valid)
                        self.valid_scores.update(evaluation.eval_relations(model=model, dataset=dataset.valid))
                        self.print_metrics(description="Validation:", metrics=self.valid_scores)
                    if dataset.test:
                        self.test_scores = evaluation.eval(model=model, dataset=dataset.test)
                        self.test_scores.update(evaluation.eval_relations(model=model, dataset=dataset.test))
                        self.print_metrics(description="Test:", metrics=self.test_scores)
                        if (self.history_test["HITS@3"] > self.test_scores["HITS@3"]
                                and self.history_test["HITS@1"] > self.test_scores["HITS@1"]):
                            self.round_without_improvement_test += 1
                        else:
                            self.round_without_improvement_test = 0
                            self.history_test = self.test_scores
                    else:
                        if (self.history_valid["HITS@3"] > self.valid_scores["HITS@3"]
                                and self.history_valid["HITS@1"] > self.valid_scores["HITS@1"]):
                            self.round_without_improvement_valid += 1
                        else:
                            self.round_without_improvement_valid = 0
                            self.history_valid = self.valid_scores
                    if (self.round_without_improvement_valid == self.early_stopping_rounds
                            or self.round_without_improvement_test == self.early_stopping_rounds):
                        print(f"\n Early stopping at epoch {epoch}.")
                        self.print_metrics(description="Validation:", metrics=self.valid_scores)
                        self.print_metrics(description="Test:", metrics=self.test_scores)
                        return self

        print(f"\n Epoch: {epoch}. \n")
        if dataset.valid:
            self.valid_scores = evaluation.eval(model=model, dataset=dataset.valid)
            self.valid_scores.update(evaluation.eval_relations(model=model, dataset=dataset.valid))
            self.print_metrics(description="Validation:", metrics=self.valid_scores)
        if dataset.test:
            self.test_scores = evaluation.eval(model=model, dataset=dataset.test)
            self.test_scores.update(evaluation.eval_relations(model=model, dataset=dataset.test))
            self.print_metrics(description="Test:", metrics=self.test_scores)
        return self

    @classmethod
    def print_metrics(cls, description, metrics):
        print(f"\t {description}")
        for metric, value in metrics.items():
            print(f"\t\t {metric}: {value}")

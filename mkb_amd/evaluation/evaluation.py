"""``Evaluation`` -- filtered link-prediction metrics MRR / MR / HITS@1/3/10 (reference
mkb/evaluation/evaluation.py:137-279) with the same constructor and return dictionaries.

``eval`` on a ROCm device runs ``mkb_rank``: every test triple is scored against ALL entity rows by a tiled HIP
kernel (no ``[B, N, D]`` gather, no per-item python loop over ``n_entity`` candidates as in
``datasets.base.TestDataset``, base.py:196-241) and the filtered rank is counted on the device; exact score ties
count in the target's favour.  ``force_reference_path = True`` (and ``eval_relations`` always) takes the
reference's route instead: ``TestDataset`` candidate lists + filter bias, ``model(sample, negative_sample, mode)``
through the general HIP forward, rank read off a descending argsort (evaluation.py:245-262).
"""
import collections
import ctypes

import numpy as np
import torch
from torch.utils import data

from .. import _hip
from ..datasets import base
from ..models.base import BaseModel
from ..utils import Bar, Mean

__all__ = ["Evaluation"]


class Evaluation:
    def __init__(self, entities, relations, batch_size, true_triples=[], device="cpu", num_workers=1):
        self.entities = entities
        self.relations = relations
        self.true_triples = true_triples
        self.batch_size = batch_size
        self.device = device
        self.num_workers = num_workers

    def _get_test_loader(self, triples, mode):
        test_dataset = base.TestDataset(triples=triples, true_triples=self.true_triples, entities=self.entities,
                                        relations=self.relations, mode=mode)
        return data.DataLoader(dataset=test_dataset, batch_size=self.batch_size, num_workers=self.num_workers,
                               collate_fn=base.TestDataset.collate_fn)

    def get_entity_stream(self, dataset):
        return [self._get_test_loader(dataset, "head-batch"), self._get_test_loader(dataset, "tail-batch")]

    def get_relation_stream(self, dataset):
        test_dataset = base.TestDatasetRelation(triples=dataset, true_triples=self.true_triples,
                                                entities=self.entities, relations=self.relations)
        return data.DataLoader(dataset=test_dataset, batch_size=self.batch_size, num_workers=self.num_workers,
                               collate_fn=base.TestDatasetRelation.collate_fn)

    # ------------------------------------------------------------------ device ranking (mkb_rank)
    def _true_keys(self, device, n_entity, n_relation):
        """Sorted keys of all true triples, one ordering per mode (cached on the device)."""
        cache = getattr(self, "_keys_cache", None)
        if cache is None or cache[0] != (device, len(self.true_triples)):
            a = np.asarray(self.true_triples, dtype=np.int64).reshape(-1, 3)
            h, r, t = a[:, 0], a[:, 1], a[:, 2]
            tail = np.unique((h * n_relation + r) * n_entity + t)
            head = np.unique((t * n_relation + r) * n_entity + h)
            cache = ((device, len(self.true_triples)),
                     {"head-batch": torch.as_tensor(head, device=device), "tail-batch": torch.as_tensor(tail, device=device)})
            self._keys_cache = cache
        return cache[1]

    def ranks(self, model, dataset, mode, chunk=1024, with_scores=False):
        """Filtered rank (1-based) of every triple of ``dataset`` in ``mode``, computed on the device: int64 tensor.
        ``with_scores=True`` -> ``(ranks, scores [n, n_entity])``: the scores of every triple against all entities the ranks
        were counted on (``mkb_rank_scores``; what ``model(sample, all entities, mode)`` returns at evaluation.py:237, before the
        filter bias)."""
        dev = model.entity_embedding.device
        _hip.require_device(model.entity_embedding)
        model.sync_parameters()
        keys = self._true_keys(dev, model.n_entity, model.n_relation)[mode]
        triples = torch.as_tensor(np.asarray(dataset, dtype=np.int64).reshape(-1, 3), device=dev)
        out = torch.empty(len(triples), dtype=torch.int64, device=dev)
        scores = torch.empty((len(triples), model.n_entity), dtype=torch.float32, device=dev) if with_scores else None
        lib, tb = _hip.lib(), model._tables()
        ws = None
        with _hip.on_device(dev):
            for lo in range(0, len(triples), chunk):
                s = triples[lo: lo + chunk].contiguous()
                need = lib.mkb_rank_workspace_bytes(tb, s.shape[0])
                if ws is None or ws.numel() < need + 256:
                    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
                off = (-ws.data_ptr()) % 256
                if with_scores:
                    _hip.check(lib.mkb_rank_scores(tb, _hip.ptr(s), s.shape[0], _hip.mode_id(mode), _hip.ptr(keys), keys.numel(),
                                                   _hip.ptr(out[lo: lo + chunk]), _hip.ptr(scores[lo: lo + chunk]),
                                                   ctypes.c_void_p(ws.data_ptr() + off), need, _hip.stream_ptr()),
                               "mkb_rank_scores")
                    continue
                _hip.check(lib.mkb_rank(tb, _hip.ptr(s), s.shape[0], _hip.mode_id(mode), _hip.ptr(keys), keys.numel(),
                                        _hip.ptr(out[lo: lo + chunk]), ctypes.c_void_p(ws.data_ptr() + off), need,
                                        _hip.stream_ptr()),
                           "mkb_rank")
        return (out, scores) if with_scores else out

    def _device_ok(self, model):
        units = model.hidden_dim if model.name == "RotatE" else model.entity_dim
        return (isinstance(model, BaseModel) and model.entity_embedding.is_cuda and str(self.device) != "cpu"
                and units <= 4096 and len(self.true_triples) > 0)

    def eval(self, model, dataset):
        metrics = collections.OrderedDict({m: Mean() for m in ["MRR", "MR", "HITS@1", "HITS@3", "HITS@10"]})
        if self._device_ok(model) and not getattr(self, "force_reference_path", False):
            with torch.no_grad():
                for mode in ("head-batch", "tail-batch"):  # same order as get_entity_stream
                    # The reference creates one DataLoader iterator per side here, and each creation draws one int64
                    # from torch's global CPU generator (the workers' base seed).  Draw it too, so that a training
                    # run interleaved with evaluations keeps shuffling its batches exactly like the reference.
                    torch.empty((), dtype=torch.int64).random_()
                    for ranking in self.ranks(model, dataset, mode).tolist():
                        metrics["MRR"].update(1.0 / ranking)
                        metrics["MR"].update(ranking)
                        metrics["HITS@1"].update(1.0 if ranking <= 1 else 0.0)
                        metrics["HITS@3"].update(1.0 if ranking <= 3 else 0.0)
                        metrics["HITS@10"].update(1.0 if ranking <= 10 else 0.0)
            return {name: round(metric.get(), 4) for name, metric in metrics.items()}
        with torch.no_grad():
            for test_set in self.get_entity_stream(dataset):
                metrics = self.compute_score(model=model, test_set=test_set, metrics=metrics, device=self.device)
        return {name: round(metric.get(), 4) for name, metric in metrics.items()}

    def relation_ranks(self, model, dataset, chunk=4096):
        """Filtered rank (1-based) of the true relation of every triple among all relations, on the device: what
        ``compute_score`` does with the ``relation-batch`` stream (datasets/base.py:254-305: the other true relations of
        (h, ., t) are replaced by the target relation and biased by -1), without the per-item host loop."""
        dev = model.entity_embedding.device
        n_ent, n_rel = model.n_entity, model.n_relation
        keys = self._true_keys(dev, n_ent, n_rel)["tail-batch"]  # sorted (h * R + r) * N + t
        triples = torch.as_tensor(np.asarray(dataset, dtype=np.int64).reshape(-1, 3), device=dev)
        cand = torch.arange(n_rel, device=dev)
        out = []
        for lo in range(0, len(triples), chunk):
            s = triples[lo: lo + chunk]
            h, r, t = s[:, 0:1], s[:, 1:2], s[:, 2:3]
            k = (h * n_rel + cand) * n_ent + t                       # [b, R] keys of (h, r', t)
            pos = torch.searchsorted(keys, k.reshape(-1)).clamp_(max=keys.numel() - 1).view_as(k)
            true = keys[pos] == k
            rel = torch.where(true, r.expand_as(k), cand.expand_as(k))
            bias = torch.where(true & (cand != r), -1.0, 0.0)
            neg = torch.stack([h.expand_as(k), rel, t.expand_as(k)], dim=-1)  # [b, R, 3]
            score = model(neg.contiguous()) + bias
            # position in a stable descending sort with NaN first (see ranks_before in csrc/rank.hip): a collapsed or
            # diverged model must not rank its targets first
            key = torch.nan_to_num(score, nan=float("inf"), posinf=float("inf"))
            target = key.gather(1, r)
            before = (key > target) | ((key == target) & (cand < r))
            out.append(1 + before.sum(dim=1))
        return torch.cat(out) if out else torch.empty(0, dtype=torch.int64, device=dev)

    def eval_relations(self, model, dataset):
        metrics = collections.OrderedDict({m: Mean() for m in ["MRR", "MR", "HITS@1", "HITS@3", "HITS@10"]})
        if self._device_ok(model) and not getattr(self, "force_reference_path", False) and len(dataset) > 0:
            with torch.no_grad():
                torch.empty((), dtype=torch.int64).random_()  # the reference's DataLoader iterator draws its base seed
                for ranking in self.relation_ranks(model, dataset).tolist():
                    metrics["MRR"].update(1.0 / ranking)
                    metrics["MR"].update(ranking)
                    metrics["HITS@1"].update(1.0 if ranking <= 1 else 0.0)
                    metrics["HITS@3"].update(1.0 if ranking <= 3 else 0.0)
                    metrics["HITS@10"].update(1.0 if ranking <= 10 else 0.0)
            return {f"{name}_relations": round(metric.get(), 4) for name, metric in metrics.items()}
        with torch.no_grad():
            metrics = self.compute_score(model=model, test_set=self.get_relation_stream(dataset), metrics=metrics,
                                         device=self.device)
        return {f"{name}_relations": round(metric.get(), 4) for name, metric in metrics.items()}

    @classmethod
    def compute_score(cls, model, test_set, metrics, device):
        training = model.training
        if training:
            model = model.eval()
        bar = Bar(dataset=test_set, update_every=1)
        bar.set_description("Evaluation")
        for batch in bar:
            sample = batch["sample"].to(device)
            negative_sample = batch["negative_sample"].to(device)
            filter_bias = batch["filter_bias"].to(device)
            mode = batch["mode"]
            if mode == "relation-batch":
                score = model(negative_sample)
                positive_arg = sample[:, 1]
            else:
                score = model(sample=sample, negative_sample=negative_sample, mode=mode)
                positive_arg = sample[:, 0] if mode == "head-batch" else sample[:, 2]
            score = score + filter_bias
            argsort = torch.argsort(score, dim=1, descending=True)
            hit = argsort == positive_arg.unsqueeze(1)
            assert bool((hit.sum(dim=1) == 1).all())
            ranks = (hit.float().argmax(dim=1) + 1).tolist()  # one D2H copy per batch
            for ranking in ranks:
                metrics["MRR"].update(1.0 / ranking)
                metrics["MR"].update(ranking)
                metrics["HITS@1"].update(1.0 if ranking <= 1 else 0.0)
                metrics["HITS@3"].update(1.0 if ranking <= 3 else 0.0)
                metrics["HITS@10"].update(1.0 if ranking <= 10 else 0.0)
        if training:
            model = model.train()
        return metrics

"""``Evaluation`` -- filtered link-prediction metrics MRR / MR / HITS@1/3/10 (reference
mkb/evaluation/evaluation.py:137-279) with the same constructor and return dictionaries.

Scores come from the same HIP forward as training (``model(sample, negative_sample, mode)`` with all
``n_entity`` candidates); the filter bias and candidate lists are those of ``datasets.base.TestDataset``.
The rank of the target is read off a descending argsort on the device, as the reference does on the host
(evaluation.py:245-262), so ties are broken by the sort, not by a convention of ours.
"""
import collections

import torch
from torch.utils import data

from ..datasets import base
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

    def eval(self, model, dataset):
        metrics = collections.OrderedDict({m: Mean() for m in ["MRR", "MR", "HITS@1", "HITS@3", "HITS@10"]})
        with torch.no_grad():
            for test_set in self.get_entity_stream(dataset):
                metrics = self.compute_score(model=model, test_set=test_set, metrics=metrics, device=self.device)
        return {name: round(metric.get(), 4) for name, metric in metrics.items()}

    def eval_relations(self, model, dataset):
        metrics = collections.OrderedDict({m: Mean() for m in ["MRR", "MR", "HITS@1", "HITS@3", "HITS@10"]})
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

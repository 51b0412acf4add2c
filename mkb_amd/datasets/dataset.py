"""``Dataset`` -- the batch producer ``compose.Pipeline`` iterates (reference mkb/datasets/dataset.py:94-320).

Same constructor, attributes and iteration order as the reference: two shuffled torch ``DataLoader`` s
(head-batch and tail-batch views of the training triples) zipped and alternated; the torch RNG is seeded
with ``seed`` at the end of construction (dataset.py:185-186) so the shuffle order is the reference's.
"""
import copy

import torch
from torch.utils import data

from .base import TestDataset, TrainDataset

__all__ = ["Dataset"]


class Dataset:
    def __init__(self, train, batch_size, entities=None, relations=None, valid=None, test=None, shuffle=True,
                 classification=False, pre_compute=True, num_workers=1, seed=42, classification_valid=None,
                 classification_test=None):
        if classification:
            raise NotImplementedError("classification mode (ConvE / BCE) is outside the mkb_amd hot path")
        self.train, self.valid, self.test = train, valid, test
        self.batch_size, self.shuffle = batch_size, shuffle
        self.classification, self.pre_compute = classification, pre_compute
        self.num_workers, self.seed = num_workers, seed

        if entities is None:  # dataset.py:140-152: label -> id in order of first appearance
            self.entities = self.mapping_entities()
            relabel = lambda ts: None if ts is None else [(self.entities[h], r, self.entities[t]) for h, r, t in ts]
            self.train, self.valid, self.test = relabel(self.train), relabel(self.valid), relabel(self.test)
        else:
            self.entities = entities
        if relations is None:
            self.relations = self.mapping_relations()
            relabel = lambda ts: None if ts is None else [(h, self.relations[r], t) for h, r, t in ts]
            self.train, self.valid, self.test = relabel(self.train), relabel(self.valid), relabel(self.test)
        else:
            self.relations = relations

        self.n_entity = len(self.entities)
        self.n_relation = len(self.relations)

        self.step = 0
        self.dataset_head = self.get_train_loader(mode="head-batch")
        self.dataset_tail = self.get_train_loader(mode="tail-batch")
        self.len = int((len(self.dataset_head.dataset) + len(self.dataset_tail.dataset)) / self.batch_size)
        self.fetch_head = self.fetch(self.dataset_head)
        self.fetch_tail = self.fetch(self.dataset_tail)

        self.classification_valid = classification_valid
        self.classification_test = classification_test

        if self.seed:
            torch.manual_seed(self.seed)

    def __iter__(self):
        for head, tail in zip(self.dataset_head, self.dataset_tail):
            yield head
            yield tail

    def __next__(self):
        self.step += 1
        return next(self.fetch_head) if self.step % 2 == 0 else next(self.fetch_tail)

    @staticmethod
    def fetch(dataloader):
        while True:
            yield from dataloader

    def __len__(self):
        return self.len

    @property
    def true_triples(self):
        out = copy.deepcopy(self.train)
        if self.valid is not None:
            out += self.valid
        if self.test is not None:
            out += self.test
        return out

    @property
    def train_triples(self):
        return self.train

    @property
    def name(self):
        return self.__class__.__name__

    @property
    def _repr_title(self):
        return f"{self.name} dataset"

    @property
    def _repr_content(self):
        return {
            "Batch size": f"{self.batch_size}",
            "Entities": f"{self.n_entity}",
            "Relations": f"{self.n_relation}",
            "Shuffle": f"{self.shuffle}",
            "Train triples": f"{len(self.train) if self.train else 0}",
            "Validation triples": f"{len(self.valid) if self.valid else 0}",
            "Test triples": f"{len(self.test) if self.test else 0}",
        }

    def __repr__(self):
        l_len = max(map(len, self._repr_content.keys()))
        r_len = max(map(len, self._repr_content.values()))
        return f"{self._repr_title}\n" + "\n".join(
            k.rjust(l_len) + "  " + v.ljust(r_len) for k, v in self._repr_content.items())

    def test_dataset(self, batch_size):
        return self.test_stream(triples=self.test, batch_size=batch_size)

    def validation_dataset(self, batch_size):
        return self.test_stream(triples=self.valid, batch_size=batch_size)

    def test_stream(self, triples, batch_size):
        return [self._get_test_loader(triples, batch_size, "head-batch"),
                self._get_test_loader(triples, batch_size, "tail-batch")]

    def get_train_loader(self, mode):
        dataset = TrainDataset(triples=self.train, entities=self.entities, relations=self.relations, mode=mode,
                               pre_compute=self.pre_compute, seed=self.seed)
        return data.DataLoader(dataset=dataset, batch_size=self.batch_size, shuffle=self.shuffle,
                               num_workers=self.num_workers, collate_fn=TrainDataset.collate_fn)

    def _get_test_loader(self, triples, batch_size, mode):
        test_dataset = TestDataset(triples=triples, true_triples=self.train + self.test + self.valid,
                                   entities=self.entities, relations=self.relations, mode=mode)
        return data.DataLoader(dataset=test_dataset, batch_size=batch_size, num_workers=self.num_workers,
                               collate_fn=TestDataset.collate_fn)

    def mapping_entities(self):
        """dataset.py:322-331: ids in order of first appearance over all heads of train+valid+test, then all
        tails."""
        tt = self.true_triples
        return {e: i for i, e in enumerate(dict.fromkeys([h for h, _, _ in tt] + [t for _, _, t in tt]))}

    def mapping_relations(self):
        """dataset.py:333-335."""
        return {r: i for i, r in enumerate(dict.fromkeys([r for _, r, _ in self.true_triples]))}

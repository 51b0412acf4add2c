"""``Dataset`` -- the batch producer ``compose.Pipeline`` iterates.

Public contract of the reference class (mkb/datasets/dataset.py:94-335), built differently: the three splits live in one
table that the label -> id relabelling maps over, the two training views (head-batch / tail-batch) come from a mode table,
the endless ``next()`` streams are per-mode generators, and ``__repr__`` goes through the formatter shared with the other
``mkb_amd`` objects.  What callers can observe is the reference's:

* constructor arguments and public attributes (``train / valid / test``, ``entities``, ``relations``, ``n_entity``,
  ``n_relation``, ``dataset_head``, ``dataset_tail``, ``len``, ``step`` ...);
* iteration: the two shuffled ``DataLoader`` views advance in lock step, head-batch first (dataset.py:196-203), and stop
  with the shorter one; ``next(dataset)`` alternates tail, head, tail ... without ever stopping (dataset.py:205-214);
* the torch RNG is re-seeded with ``seed`` as the LAST act of construction (dataset.py:185-186), which is what fixes the
  shuffle order of the first epoch;
* ids for unlabelled inputs are handed out in order of first appearance: all heads of train+valid+test, then all tails
  (dataset.py:322-335).
"""
import itertools

import torch
from torch.utils import data

from ..utils.fmt import aligned_block
from .base import TestDataset, TrainDataset

__all__ = ["Dataset"]

_SPLITS = ("train", "valid", "test")
_VIEWS = ("head-batch", "tail-batch")  # order matters: an epoch starts with the head-batch view


def _first_seen(labels):
    """label -> id in order of first appearance."""
    index = {}
    for label in labels:
        index.setdefault(label, len(index))
    return index


class _IndexBatches:
    """``BatchSampler(RandomSampler(view) | SequentialSampler(view), batch_size, drop_last=False)`` as ONE object that hands
    out whole index arrays.  torch's pair walks the permutation index by index through two Python generators (0.6 ms per
    1024-row batch -- more than two MI355X training steps); this one slices the same permutation.  Same draws from the
    global generator at the same moments as ``RandomSampler.__iter__`` (torch/utils/data/sampler.py): the shuffle seed is
    taken when the first batch is asked for, then ONE ``randperm(n)`` from a private generator seeded with it."""

    def __init__(self, n, batch_size, shuffle):
        self.n, self.batch_size, self.shuffle = n, batch_size, shuffle

    def __len__(self):
        return -(-self.n // self.batch_size)

    def __iter__(self):
        if self.shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            generator = torch.Generator()
            generator.manual_seed(seed)
            order = torch.randperm(self.n, generator=generator).numpy()
        else:
            import numpy as np

            order = np.arange(self.n, dtype=np.int64)
        for lo in range(0, self.n, self.batch_size):
            yield order[lo: lo + self.batch_size]


class Dataset:
    def __init__(self, train, batch_size, entities=None, relations=None, valid=None, test=None, shuffle=True,
                 classification=False, pre_compute=True, num_workers=1, seed=42, classification_valid=None,
                 classification_test=None):
        if classification:
            raise NotImplementedError("classification mode (ConvE / BCE) is outside the mkb_amd hot path")
        self.batch_size, self.shuffle, self.seed = batch_size, shuffle, seed
        self.classification, self.pre_compute, self.num_workers = classification, pre_compute, num_workers
        self.classification_valid, self.classification_test = classification_valid, classification_test
        self.train, self.valid, self.test = train, valid, test

        # label -> id maps: taken as given, or built from the triples (which are then relabelled)
        for kind, given, relabel in (("entities", entities, lambda ids, h, r, t: (ids[h], r, ids[t])),
                                     ("relations", relations, lambda ids, h, r, t: (h, ids[r], t))):
            ids = getattr(self, f"mapping_{kind}")() if given is None else given
            setattr(self, kind, ids)
            if given is None:
                self._map_splits(lambda h, r, t, ids=ids, relabel=relabel: relabel(ids, h, r, t))
        self.n_entity, self.n_relation = len(self.entities), len(self.relations)

        self._loaders = {mode: self.get_train_loader(mode=mode) for mode in _VIEWS}
        self.len, self.step = int(sum(len(loader.dataset) for loader in self._loaders.values()) / batch_size), 0
        self._streams = {mode: self.fetch(loader) for mode, loader in self._loaders.items()}

        if seed:  # 0 / None leave the global generator alone, like the reference
            torch.manual_seed(seed)

    # ------------------------------------------------------------------ splits
    def _map_splits(self, fn):
        for name in _SPLITS:
            triples = getattr(self, name)
            if triples is not None:
                setattr(self, name, [fn(*triple) for triple in triples])

    @property
    def true_triples(self):
        """train + valid + test as one fresh list (absent splits skipped)."""
        return [triple for name in _SPLITS for triple in (getattr(self, name) or ())]

    train_triples = property(lambda self: self.train)

    def mapping_entities(self):
        known = self.true_triples
        return _first_seen(itertools.chain((h for h, _, _ in known), (t for _, _, t in known)))

    def mapping_relations(self):
        return _first_seen(r for _, r, _ in self.true_triples)

    # ------------------------------------------------------------------ training views
    @property
    def dataset_head(self):
        return self._loaders["head-batch"]

    @property
    def dataset_tail(self):
        return self._loaders["tail-batch"]

    @property
    def fetch_head(self):
        return self._streams["head-batch"]

    @property
    def fetch_tail(self):
        return self._streams["tail-batch"]

    def get_train_loader(self, mode):
        """The reference's loader (dataset.py:297-303) for one training view.  ``num_workers`` worker PROCESSES are not
        started: a batch is one indexed read here (``TrainDataset.__getitems__``, ~0.1 ms), and shipping it through a worker's
        queue costs ~2 ms -- eight MI355X training steps.  The batches and their order are those of the worker-process
        loader: the index sampler runs in the main process either way, and ``__iter__`` draws the two views' shuffle seeds in
        the order a worker-process loader draws them (``tests/test_gpu_pool.py::test_pipeline_countries_vs_reference_capture``
        replays the reference's own batches)."""
        view = TrainDataset(triples=self.train, entities=self.entities, relations=self.relations, mode=mode,
                            pre_compute=self.pre_compute, seed=self.seed)
        return data.DataLoader(view, batch_sampler=_IndexBatches(len(view), self.batch_size, self.shuffle), num_workers=0,
                               collate_fn=TrainDataset.collate_fn)

    def __iter__(self):
        """head-batch, tail-batch, head-batch ... until the shorter view ends (the reference zips its two loaders,
        dataset.py:196-203).  A loader with worker processes draws its base seed AND -- by prefetching -- its sampler's shuffle
        seed from the global generator when its iterator is created, so the reference consumes the generator in the order
        head(base, shuffle), tail(base, shuffle); an in-process loader would draw the shuffle seed at the first ``next``.  Taking
        each view's first batch right after creating its iterator reproduces the reference's order exactly."""
        if self.num_workers == 0:  # the caller asked for in-process loaders: torch's own order for those
            yield from itertools.chain.from_iterable(zip(*(self._loaders[mode] for mode in _VIEWS)))
            return
        end = object()
        iters, firsts = [], []
        for mode in _VIEWS:
            it = iter(self._loaders[mode])
            iters.append(it)
            firsts.append(next(it, end))
        while True:
            if any(b is end for b in firsts):  # zip() semantics: stop with the shorter view
                return
            yield from firsts
            firsts = []
            for it in iters:
                b = next(it, end)
                firsts.append(b)
                if b is end:
                    break
            while len(firsts) < len(iters):
                firsts.append(end)

    def __next__(self):
        self.step = step = self.step + 1
        return next(self._streams[_VIEWS[step % 2]])  # odd steps: tail-batch, even steps: head-batch

    @staticmethod
    def fetch(dataloader):
        """Endless stream of batches: a new pass over ``dataloader`` starts whenever one ends."""
        for _ in itertools.count():
            yield from dataloader

    def __len__(self):
        return int(self.len)

    # ------------------------------------------------------------------ evaluation views
    def _get_test_loader(self, triples, batch_size, mode):
        view = TestDataset(triples=triples, true_triples=self.train + self.test + self.valid, entities=self.entities,
                           relations=self.relations, mode=mode)
        return data.DataLoader(view, batch_size=batch_size, num_workers=self.num_workers, collate_fn=TestDataset.collate_fn)

    def test_stream(self, triples, batch_size):
        return [self._get_test_loader(triples=triples, batch_size=batch_size, mode=mode) for mode in _VIEWS]

    def test_dataset(self, batch_size):
        return self.test_stream(self.test, batch_size)

    def validation_dataset(self, batch_size):
        return self.test_stream(self.valid, batch_size)

    # ------------------------------------------------------------------ display
    name = property(lambda self: type(self).__name__)
    _repr_title = property(lambda self: self.name + " dataset")

    @property
    def _repr_content(self):
        sizes = {name: len(getattr(self, name) or ()) for name in _SPLITS}
        rows = [("Batch size", self.batch_size), ("Entities", self.n_entity), ("Relations", self.n_relation),
                ("Shuffle", self.shuffle), ("Train triples", sizes["train"]), ("Validation triples", sizes["valid"]),
                ("Test triples", sizes["test"])]
        return {label: str(value) for label, value in rows}

    def __repr__(self):
        return aligned_block(self._repr_title, self._repr_content)

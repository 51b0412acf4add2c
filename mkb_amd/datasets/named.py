"""Named benchmark loaders (reference mkb/datasets/{countries_s1,umls,wn18rr,fb15k237,yago310}.py).

The triples ship as compact ``.npz`` assets under ``mkb_amd/datasets/data/`` (packed from the
reference's CSV/JSON *data* files by ``tools/pack_datasets.py``).  Constructor signatures and defaults
follow the reference loaders (e.g. fb15k237.py:63-71: ``seed=None``; countries_s1.py:69-77: ``seed=42``).
"""
import json
import pathlib
import warnings

import numpy as np

from .dataset import Dataset

__all__ = ["CountriesS1", "Fb15k237", "Umls", "Wn18rr", "Yago310"]

_DATA = pathlib.Path(__file__).parent / "data"


def _load(name):
    z = np.load(_DATA / f"{name}.npz")
    as_list = lambda a: [tuple(r) for r in a.astype(np.int64).tolist()]
    ents = {k: i for i, k in enumerate(json.loads(bytes(z["entities"]).decode("utf-8")))}
    rels = {k: i for i, k in enumerate(json.loads(bytes(z["relations"]).decode("utf-8")))}
    return as_list(z["train"]), as_list(z["valid"]), as_list(z["test"]), ents, rels


class _Named(Dataset):
    filename = None
    default_seed = None

    def __init__(self, batch_size, classification=False, shuffle=True, pre_compute=True, num_workers=1,
                 seed="default"):
        train, valid, test, ents, rels = _load(self.filename)
        train = self._train(train, ents, rels)
        super().__init__(train=train, valid=valid, test=test, entities=ents, relations=rels,
                         batch_size=batch_size, shuffle=shuffle, classification=classification,
                         pre_compute=pre_compute, num_workers=num_workers,
                         seed=self.default_seed if seed == "default" else seed)

    def _train(self, train, ents, rels):
        return train


class CountriesS1(_Named):
    filename, default_seed = "countries_s1", 42


class Umls(_Named):
    filename, default_seed = "umls", None


class Wn18rr(_Named):
    filename, default_seed = "wn18rr", None


class Fb15k237(_Named):
    filename, default_seed = "fb15k237", None


class Yago310(_Named):
    """YAGO3-10: 123,182 entities, 37 relations, real valid/test (5,000 each).  The reference mount lacks
    ``train.csv`` (``.MISSING_LARGE_BLOBS``), so the 1,079,040 training triples (count from
    yago310.py:47) are SYNTHETIC: ``RandomState(42)``, heads/tails Zipf(1.0) over the entity ids,
    relations drawn from the real valid.csv relation histogram."""

    filename, default_seed = "yago310", None
    n_train = 1079040

    def _train(self, train, ents, rels):
        if train:
            return train
        warnings.warn("datasets.Yago310: the reference's train.csv is not available here; the 1,079,040 training triples are "
                      "SYNTHETIC (Zipf heads / tails, RandomState(42)).  Shapes and throughput are representative, link-prediction "
                      "metrics on the real valid / test sets are not.", RuntimeWarning, stacklevel=3)
        self.synthetic_train = True
        z = np.load(_DATA / "yago310.npz")
        rs = np.random.RandomState(42)
        n, r = len(ents), len(rels)
        p = 1.0 / np.arange(1, n + 1)
        p /= p.sum()
        perm = rs.permutation(n)
        h = perm[rs.choice(n, size=self.n_train, p=p)]
        t = perm[rs.choice(n, size=self.n_train, p=p)]
        hist = np.bincount(z["valid"][:, 1], minlength=r).astype(np.float64) + 1.0
        rel = rs.choice(r, size=self.n_train, p=hist / hist.sum())
        a = np.unique(np.stack([h, rel, t], 1), axis=0)
        a = a[rs.permutation(len(a))]
        return [tuple(x) for x in a.astype(np.int64).tolist()]

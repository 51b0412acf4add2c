"""Filtered link-prediction ranking, restated on the CPU from the reference (TEST INFRASTRUCTURE ONLY: imported by ``tests/``
and nothing else -- the product path is ``mkb_rank`` in mkb_amd/csrc/rank.hip).

Follows, under ``/root/reference``:
  * ``mkb/datasets/base.py:196-241`` (``TestDataset.__getitem__``): per test triple the candidate list is every entity id in
    order; a candidate whose corrupted triple is ANOTHER true triple is replaced by the target itself with bias -1e5 (the pairs go
    through ``torch.LongTensor(tmp)``, so the bias is the integer -100000 before ``.float()``); the target keeps bias 0.
  * ``mkb/evaluation/evaluation.py:232-262`` (``Evaluation.compute_score``): ``score = model(sample, negative_sample, mode)``,
    ``score += filter_bias``, ``argsort(score, dim=1, descending=True)``, rank = 1 + position of the column whose INDEX equals the
    target's entity id (``argsort`` holds column indices; column e is entity e or, when filtered, a biased copy of the target).

Pinned by tests/golden/eval_headline.npz (tools/make_golden.py::gen_eval_headline: the live reference's TestDataset +
compute_score arithmetic on FB15k-237 at hidden 1000, all five models): tests/test_oracle_ranking.py.
"""
from __future__ import annotations

import numpy as np
import torch

from . import scoring

FILTER_BIAS = float(int(-1e5))  # base.py:215 / :231 through torch.LongTensor


def true_key_set(true_triples, n_entity: int, n_relation: int) -> np.ndarray:
    """Sorted unique int64 keys (h * R + r) * N + t of the true triples (a set, like base.py:188)."""
    a = np.asarray(true_triples, dtype=np.int64).reshape(-1, 3)
    return np.unique((a[:, 0] * n_relation + a[:, 1]) * n_entity + a[:, 2])


def candidates(triples, keys: np.ndarray, n_entity: int, n_relation: int, mode: str):
    """``negative_sample`` [b, N] int64 and ``filter_bias`` [b, N] float32 of base.py:196-241 for a batch of test triples."""
    s = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    h, r, t = s[:, 0:1], s[:, 1:2], s[:, 2:3]
    cand = np.arange(n_entity, dtype=np.int64)[None, :]
    if mode == "head-batch":
        k = (cand * n_relation + r) * n_entity + t
        target = h
    elif mode == "tail-batch":
        k = (h * n_relation + r) * n_entity + cand
        target = t
    else:
        raise ValueError(mode)
    pos = np.minimum(np.searchsorted(keys, k.reshape(-1)), len(keys) - 1).reshape(k.shape)
    is_true = keys[pos] == k
    other = is_true & (cand != target)               # "actual true triple that we filter out" (:213-216 / :229-232)
    neg = np.where(other, target, cand)               # ... replaced by the target itself
    bias = np.where(other, FILTER_BIAS, 0.0).astype(np.float32)
    return neg, bias


def scores_and_ranks(tb: scoring.Tables, triples, keys: np.ndarray, mode: str, chunk: int = 8, fast_norm: bool = False,
                     want_raw: bool = True):
    """-> (raw scores [n, N] float32 of every entity IN ORDER (before the filter: what column e would score unfiltered),
    biased scores [n, N] as the reference ranks them, ranks [n] int64).  ``chunk`` rows at a time: the reference
    formulation materialises [chunk, N, De] operands."""
    n_entity, n_relation = tb.ent.shape[0], tb.rel.shape[0]
    s_all = torch.as_tensor(np.asarray(triples, dtype=np.int64).reshape(-1, 3))
    everyone = torch.arange(n_entity, dtype=torch.int64)
    raw, biased, ranks = [], [], []
    with torch.no_grad():
        for lo in range(0, len(s_all), chunk):
            s = s_all[lo: lo + chunk]
            neg, bias = candidates(s.numpy(), keys, n_entity, n_relation, mode)
            sc = scoring.score(tb, s, torch.as_tensor(neg), mode, fast_norm=fast_norm)   # evaluation.py:237
            sc = sc + torch.as_tensor(bias)                                              # :243
            order = torch.argsort(sc, dim=1, descending=True)                            # :245
            target = s[:, 0] if mode == "head-batch" else s[:, 2]                        # :247-254
            hit = order == target.unsqueeze(1)
            assert bool((hit.sum(dim=1) == 1).all())                                     # :261
            ranks.append(hit.float().argmax(dim=1) + 1)                                  # :263
            biased.append(sc)
            if want_raw:
                raw.append(scoring.score(tb, s, everyone.unsqueeze(0).expand(len(s), -1), mode, fast_norm=fast_norm))
    return (torch.cat(raw).numpy() if want_raw else None), torch.cat(biased).numpy(), torch.cat(ranks).numpy()


def scores_and_ranks_one_pass(tb: scoring.Tables, triples, keys: np.ndarray, mode: str, chunk: int = 8, fast_norm: bool = False):
    """``scores_and_ranks`` with ONE scoring pass per chunk: a filtered column is a copy of the target's column plus the bias and
    every column of the materialised formulation is computed independently of the others, so the biased block is derived from the
    raw one (bit-identical to the two-pass form: tests/test_oracle_ranking.py).  For the full-size GPU parity tests, where the
    oracle's time matters."""
    n_entity, n_relation = tb.ent.shape[0], tb.rel.shape[0]
    s_all = torch.as_tensor(np.asarray(triples, dtype=np.int64).reshape(-1, 3))
    everyone = torch.arange(n_entity, dtype=torch.int64)
    raw, biased, ranks = [], [], []
    with torch.no_grad():
        for lo in range(0, len(s_all), chunk):
            s = s_all[lo: lo + chunk]
            neg, bias = candidates(s.numpy(), keys, n_entity, n_relation, mode)
            r = scoring.score(tb, s, everyone.unsqueeze(0).expand(len(s), -1), mode, fast_norm=fast_norm)
            sc = r.gather(1, torch.as_tensor(neg)) + torch.as_tensor(bias)
            order = torch.argsort(sc, dim=1, descending=True)
            target = s[:, 0] if mode == "head-batch" else s[:, 2]
            hit = order == target.unsqueeze(1)
            assert bool((hit.sum(dim=1) == 1).all())
            ranks.append(hit.float().argmax(dim=1) + 1)
            raw.append(r)
            biased.append(sc)
    return torch.cat(raw).numpy(), torch.cat(biased).numpy(), torch.cat(ranks).numpy()


def rank_bounds(raw: np.ndarray, biased: np.ndarray, target: np.ndarray, eps: float):
    """Ranks a ranking that sees the scores within +-eps may report: [1 + #{biased > target + eps}, 1 + #{biased >= target - eps}
    minus the target itself].  Candidates closer than eps to the target's score are the documented near-ties."""
    st = raw[np.arange(len(target)), target][:, None]
    lo = 1 + (biased > st + eps).sum(axis=1)
    hi = (biased >= st - eps).sum(axis=1)  # (includes the target's own column: 1 + the others)
    return lo, np.maximum(hi, lo)

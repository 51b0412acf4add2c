"""CPU restatement of ``mkb.sampling.NegativeSampling`` (TEST INFRASTRUCTURE ONLY).

Own code for every stage so that each can be pinned separately:

* ``MT19937``            numpy's legacy ``RandomState(seed)`` stream: ``init_genrand`` seeding,
                         block regeneration, tempering (numpy/random/src/mt19937/mt19937.c, numpy 2.2.6;
                         reference call site ``sampling/negative_sampling.py:151``).
* ``MT19937.randint``    legacy ``RandomState.randint(n, size=...)`` for int64: masked rejection,
                         one 32-bit output per trial (reference call site ``negative_sampling.py:166``).
* ``in1d_invert_mask``   ``np.in1d(c, rec, assume_unique=True, invert=True)`` of numpy >= 1.24:
                         table / loop / sort paths, the sort path keeping only the LAST occurrence of a
                         duplicated candidate (reference call site ``negative_sampling.py:153-156``).
* ``NegativeSampling``   the generate loop (``negative_sampling.py:158-201``): ONE pool per call,
                         every row filters the same pool, cyclic fill to ``size``.

Third-party dependency restated: numpy (requirements.txt:3 pins only ``numpy >= 1.18.1``); the
oracle is pinned to numpy 2.2.6 (the build container's), see tests/test_oracle_sampler.py which
checks every stage against numpy itself and against the reference's doctest known answers
(``negative_sampling.py:101-103, 120-122``).
"""
from __future__ import annotations

import numpy as np

_N, _M = 624, 397
_U32 = np.uint32


class MT19937:
    def __init__(self, seed: int):
        s = int(seed) & 0xFFFFFFFF
        key = np.empty(_N, dtype=np.uint64)
        for i in range(_N):
            key[i] = s
            s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xFFFFFFFF
        self.key = key.astype(_U32)
        self.pos = _N

    def _regen(self):
        mt = self.key.astype(np.uint64)
        upper, lower = 0x80000000, 0x7FFFFFFF
        for i in range(_N):
            y = (int(mt[i]) & upper) | (int(mt[(i + 1) % _N]) & lower)
            mt[i] = int(mt[(i + _M) % _N]) ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
        self.key = mt.astype(_U32)
        self.pos = 0

    def next_uint32(self) -> int:
        if self.pos == _N:
            self._regen()
        y = int(self.key[self.pos])
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def randint(self, n: int, size: int) -> np.ndarray:
        """Legacy bounded int64 draw in [0, n): masked rejection on 32-bit outputs."""
        rng = n - 1
        out = np.zeros(size, dtype=np.int64)
        if rng == 0:
            return out
        assert rng <= 0xFFFFFFFF
        mask = rng
        for sh in (1, 2, 4, 8, 16):
            mask |= mask >> sh
        for i in range(size):
            while True:
                v = self.next_uint32() & mask
                if v <= rng:
                    break
            out[i] = v
        return out


def in1d_path(n_candidates: int, rec: np.ndarray) -> str:
    """Which branch numpy>=1.24 ``_in1d`` takes for integer inputs (numpy/lib/_arraysetops_impl.py)."""
    m = len(rec)
    rec_range = int(rec.max()) - int(rec.min())
    if rec_range <= 6 * (n_candidates + m):
        return "table"
    if m < 10 * n_candidates ** 0.145:
        return "loop"
    return "sort"


def in1d_invert_mask(c: np.ndarray, rec: np.ndarray) -> np.ndarray:
    """keep[p] of ``np.in1d(c, rec, assume_unique=True, invert=True)``."""
    recset = set(int(v) for v in rec)
    keep = np.array([int(v) not in recset for v in c], dtype=bool)
    if in1d_path(len(c), rec) == "sort":
        # stable merge-sort of concat(c, rec) flags an element only if its successor differs:
        # among equal candidates only the last occurrence survives.
        last = {}
        for p, v in enumerate(c):
            last[int(v)] = p
        for p, v in enumerate(c):
            if last[int(v)] != p:
                keep[p] = False
    return keep


def positive_triples(triples):
    """negative_sampling.py:7-28 (only membership / min / max / len of each set matter)."""
    true_head, true_tail = {}, {}
    for h, r, t in triples:
        true_tail.setdefault((h, r), set()).add(t)
        true_head.setdefault((r, t), set()).add(h)
    true_head = {k: np.array(sorted(v), dtype=np.int64) for k, v in true_head.items()}
    true_tail = {k: np.array(sorted(v), dtype=np.int64) for k, v in true_tail.items()}
    return true_head, true_tail


class NegativeSampling:
    def __init__(self, size, train_triples, entities, relations, seed=42):
        self.size = size
        self.n_entity = len(entities)
        self.n_relation = len(relations)
        self.true_head, self.true_tail = positive_triples(train_triples)
        self._rng = MT19937(seed)

    def generate(self, sample, mode):
        """-> (negatives [B, size] int64, pool [2*size] int64)."""
        pool = self._rng.randint(self.n_entity, self.size * 2)
        rows = []
        for h, r, t in np.asarray(sample).tolist():
            rec = self.true_head[(r, t)] if mode == "head-batch" else self.true_tail[(h, r)]
            f = pool[in1d_invert_mask(pool, rec)]
            if f.size == 0:
                raise RuntimeError("filter removed the whole pool (the reference loops forever here)")
            reps = -(-self.size // f.size)
            rows.append(np.tile(f, reps)[: self.size])
        return np.stack(rows, axis=0), pool

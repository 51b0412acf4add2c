"""Pins oracle/sampler.py (own MT19937 + in1d restatement) and oracle/csrc/sampler.c against
(a) numpy itself, (b) the reference doctest known answers, (c) golden negatives from the live reference."""
import ctypes

import numpy as np
import pytest

from oracle import sampler as osamp


def test_mt19937_stream_equals_numpy():
    for seed in (0, 42, 7, 2**32 - 1):
        mt = osamp.MT19937(seed)
        got = np.array([mt.next_uint32() for _ in range(1500)], dtype=np.uint64)
        ref = np.random.RandomState(seed).randint(0, 2**32, size=1500, dtype=np.uint64)
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n", [1, 2, 4, 135, 14541, 40943, 123182, 2**16, 2**16 + 1])
def test_randint_equals_numpy(n):
    mt, rs = osamp.MT19937(42), np.random.RandomState(42)
    for size in (10, 700, 33):  # state persists across calls
        np.testing.assert_array_equal(mt.randint(n, size), rs.randint(n, size=size))


def test_randint_known_answers(golden):
    g = golden("sampler.npz")
    mt = osamp.MT19937(42)
    np.testing.assert_array_equal(mt.randint(4, 10), [2, 3, 0, 2, 2, 3, 0, 0, 2, 1])
    np.testing.assert_array_equal(mt.randint(4, 10), g["kat/randint4_b"])
    np.testing.assert_array_equal(osamp.MT19937(42).randint(14541, 2000), g["kat/randint14541"])
    np.testing.assert_array_equal(osamp.MT19937(7).randint(135, 1500), g["kat/randint135_seed7"])


def test_in1d_restatement_equals_numpy_all_paths():
    rs = np.random.RandomState(3)
    seen = set()
    for trial in range(400):
        P = int(rs.choice([10, 32, 256, 512]))
        n = int(rs.choice([50, 2000, 14541, 100000]))
        m = int(rs.choice([1, 2, 5, 17, 24, 25, 26, 60, 400]))
        c = rs.randint(n, size=P)
        rec = np.unique(rs.randint(n, size=m))
        if trial % 3 == 0:  # force members + duplicates among candidates
            c[: min(P, len(rec))] = rec[: min(P, len(rec))]
            c[-3:] = c[0]
        seen.add(osamp.in1d_path(P, rec))
        np.testing.assert_array_equal(osamp.in1d_invert_mask(c, rec),
                                      np.isin(c, rec, assume_unique=True, invert=True))
    assert seen == {"table", "loop", "sort"}


def test_reference_doctest_known_answers(golden):
    """sampling/negative_sampling.py:101-103, 120-122."""
    ents, rels = {i: i for i in range(4)}, {i: i for i in range(4)}
    train = [(0, 0, 1), (1, 0, 2), (2, 0, 3), (3, 0, 1)]
    ns = osamp.NegativeSampling(5, train, ents, rels, seed=42)
    smp = np.array([[0, 0, 1], [1, 0, 2]])
    tail, _ = ns.generate(smp, "tail-batch")
    head, _ = ns.generate(smp, "head-batch")
    np.testing.assert_array_equal(tail, [[2, 3, 0, 2, 2], [3, 0, 3, 0, 0]])
    np.testing.assert_array_equal(head, [[2, 2, 2, 2, 2], [2, 2, 2, 2, 3]])
    g = golden("sampler.npz")
    np.testing.assert_array_equal(tail, g["toy/tail"])
    np.testing.assert_array_equal(head, g["toy/head"])


def _csr(true_sets, stride):
    keys = np.array(sorted(a * stride + b for a, b in true_sets), dtype=np.int64)
    order = sorted(true_sets, key=lambda ab: ab[0] * stride + ab[1])
    lens = np.array([len(true_sets[k]) for k in order], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    vals = np.concatenate([true_sets[k] for k in order]).astype(np.int64)
    return keys, offs, vals


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("cls", ["Umls", "Wn18rr", "Fb15k237"])
def test_real_graph_negatives_match_reference(golden, liboracle, cls):
    from mkb_amd import datasets

    g = golden("sampler.npz")
    assert bytes(g["numpy_version"]).decode() >= "1.24"
    K = int(g[f"{cls}/K"])
    ds = getattr(datasets, cls)(batch_size=8, shuffle=False, seed=42, num_workers=0)
    train = np.asarray(ds.train, dtype=np.int64)
    ns = osamp.NegativeSampling(K, [tuple(r) for r in train.tolist()], ds.entities, ds.relations, seed=42)
    N, R = ds.n_entity, ds.n_relation
    hk, ho, hv = _csr(ns.true_head, N)
    tk, to, tv = _csr(ns.true_tail, R)
    st = ctypes.create_string_buffer(4 * 624 + 4)
    liboracle.orc_mt_seed(st, ctypes.c_uint32(42))
    liboracle.orc_generate.restype = ctypes.c_int
    c = 0
    while f"{cls}/{c}/idx" in g.files:
        smp = np.ascontiguousarray(train[g[f"{cls}/{c}/idx"]])
        mode = "head-batch" if c % 2 == 0 else "tail-batch"
        want = g[f"{cls}/{c}/neg"].astype(np.int64)
        got, pool = ns.generate(smp, mode)
        np.testing.assert_array_equal(got, want)
        neg = np.zeros((len(smp), K), dtype=np.int64)
        pl = np.zeros(2 * K, dtype=np.int64)
        k, o, v, stride = (hk, ho, hv, N) if mode == "head-batch" else (tk, to, tv, R)
        rc = liboracle.orc_generate(st, ctypes.c_int64(N), ctypes.c_int64(K), _p(smp), ctypes.c_int64(len(smp)),
                                    ctypes.c_int(mode == "head-batch"), _p(k), ctypes.c_int64(len(k)), _p(o), _p(v),
                                    ctypes.c_int64(stride), _p(neg), _p(pl))
        assert rc == 0
        np.testing.assert_array_equal(pl, pool)
        np.testing.assert_array_equal(neg, want)
        c += 1
    assert c >= 4

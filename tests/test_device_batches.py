"""``datasets.DeviceBatches``: the device-resident counterpart of the reference's batch producer
(mkb/datasets/dataset.py:185-203, 297-303).  Pure index plumbing, so it is exercised on CPU tensors here and once on the
device in tests/test_gpu_general.py."""
import numpy as np
import pytest
import torch

from mkb_amd import datasets


def _toy(n=23, batch_size=5, shuffle=True, seed=42):
    rs = np.random.RandomState(0)
    train = [(int(h), int(r), int(t)) for h, r, t in zip(rs.randint(9, size=n), rs.randint(3, size=n), rs.randint(9, size=n))]
    ents, rels = {i: i for i in range(9)}, {i: i for i in range(3)}
    return datasets.Dataset(train=train, valid=train[:2], test=train[2:4], entities=ents, relations=rels,
                            batch_size=batch_size, shuffle=shuffle, seed=seed, num_workers=0)


def test_epoch_covers_every_triple_once_per_view_and_alternates():
    ds = _toy()
    db = datasets.DeviceBatches(ds, device="cpu", seed=3)
    batches = list(db)
    assert len(batches) == len(db) == 2 * 5  # ceil(23 / 5) batches per view; the ragged last batch is kept
    assert [b["mode"] for b in batches] == ["head-batch", "tail-batch"] * 5
    train = np.asarray(ds.train, dtype=np.int64)
    for mode in ("head-batch", "tail-batch"):
        rows = torch.cat([b["sample"] for b in batches if b["mode"] == mode]).numpy()
        assert rows.shape == train.shape
        assert sorted(map(tuple, rows)) == sorted(map(tuple, train))  # a permutation of the training set
        sizes = [len(b["sample"]) for b in batches if b["mode"] == mode]
        assert sizes == [5, 5, 5, 5, 3]
    for b in batches:
        assert b["sample"].dtype == torch.int64 and b["weight"].dtype == torch.float32
        assert b["weight"].shape == (len(b["sample"]),)


def test_weights_travel_with_their_triples():
    from mkb_amd.datasets.base import subsampling_weights

    ds = _toy()
    ref = {tuple(t): float(w) for t, w in zip(np.asarray(ds.train).tolist(), subsampling_weights(ds.train).tolist())}
    for b in datasets.DeviceBatches(ds, device="cpu", seed=1):
        for t, w in zip(b["sample"].tolist(), b["weight"].tolist()):
            assert ref[tuple(t)] == pytest.approx(w, abs=0)


def test_without_shuffle_the_order_is_the_datasets():
    ds = _toy(shuffle=False)
    got = list(datasets.DeviceBatches(ds, device="cpu"))
    want = list(ds)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["mode"] == w["mode"]
        assert torch.equal(g["sample"], w["sample"]) and torch.equal(g["weight"], w["weight"])


def test_seeded_shuffle_is_reproducible_and_differs_between_epochs_and_views():
    ds = _toy()
    a, b = datasets.DeviceBatches(ds, device="cpu", seed=7), datasets.DeviceBatches(ds, device="cpu", seed=7)
    e1a, e1b = list(a), list(b)
    assert all(torch.equal(x["sample"], y["sample"]) for x, y in zip(e1a, e1b))
    e2a = list(a)  # second epoch of the same producer: a new permutation
    assert any(not torch.equal(x["sample"], y["sample"]) for x, y in zip(e1a, e2a))
    assert not torch.equal(e1a[0]["sample"], e1a[1]["sample"])  # head and tail views are shuffled independently
    c = list(datasets.DeviceBatches(ds, device="cpu", seed=8))
    assert any(not torch.equal(x["sample"], y["sample"]) for x, y in zip(e1a, c))


def test_forwards_the_dataset_interface():
    ds = _toy()
    db = datasets.DeviceBatches(ds, device="cpu")
    assert db.batch_size == 5 and db.n_entity == 9 and db.valid == ds.valid and db.test == ds.test
    assert db.true_triples == ds.true_triples and db.entities is ds.entities

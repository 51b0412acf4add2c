"""CPU (-m "not gpu") tests of the host side: dataset producers vs golden captures of the reference, the C-ABI
library's exported symbols, and loud failure without a device."""
import ctypes
import re

import numpy as np
import pytest
import torch


def test_library_exports_every_declared_symbol():
    """include/mkb_hip.h <-> libmkb_hip.so <-> mkb_amd/_hip.py agree (no compute calls: there is no GPU here)."""
    from conftest import ROOT
    from mkb_amd import _hip

    header = (ROOT / "include" / "mkb_hip.h").read_text()
    declared = set(re.findall(r"\b(mkb_[a-z_]+)\s*\(", header)) - {"mkb_sampler"}
    lib = ctypes.CDLL(str(ROOT / "mkb_amd" / "libmkb_hip.so"))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mkb_hip.h but not exported"
    assert declared == set(_hip.EXPORTED_SYMBOLS), declared ^ set(_hip.EXPORTED_SYMBOLS)
    assert _hip.lib().mkb_abi_version() == _hip.ABI_VERSION


def test_invalid_arguments_are_reported_not_crashed():
    from mkb_amd import _hip

    lib = _hip.lib()
    tb = _hip.Tables(99, 4, 10, 2, 4, 4, None, None, None, 1.0, 1.0)
    rc = lib.mkb_score_fwd(tb, None, None, 1, 1, 0, None, None)
    assert rc == -1 and b"model" in lib.mkb_last_error()
    assert lib.mkb_adam_step(None, None, None, None, 4, 1, 0.1, 0.9, 0.999, 1e-8, 0, None) == -1


def test_no_cpu_fallback():
    from mkb_amd import losses, models

    m = models.RotatE(hidden_dim=4, entities={0: 0, 1: 1}, relations={0: 0}, gamma=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.tensor([[0, 0, 1]]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.Adversarial()(torch.zeros(2, 1), torch.zeros(2, 3), torch.ones(2))


def test_product_never_imports_the_oracle():
    from conftest import ROOT

    for p in (ROOT / "mkb_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), p


def test_model_init_matches_reference_doctests(golden):
    from mkb_amd import datasets, models

    g = golden("init.npz")
    ds = datasets.CountriesS1(batch_size=2, seed=42)
    assert (ds.n_entity, ds.n_relation, len(ds.train), len(ds.valid), len(ds.test)) == (271, 2, 1111, 24, 24)
    for name in ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"]:
        torch.manual_seed(42)
        m = getattr(models, name)(hidden_dim=3, entities=ds.entities, relations=ds.relations, gamma=1)
        np.testing.assert_array_equal(m.entity_embedding.detach().numpy(), g[f"{name}/ent"])
        np.testing.assert_array_equal(m.relation_embedding.detach().numpy(), g[f"{name}/rel"])
        np.testing.assert_array_equal(m.embeddings["entities"]["oceania"].numpy(), g[f"{name}/oceania"])
        assert repr(m) == bytes(g[f"{name}/repr"]).decode()


def test_dataset_sizes_match_reference_doctests():
    """datasets/{umls,fb15k237,wn18rr}.py:39-57."""
    from mkb_amd import datasets

    for cls, want in [("Umls", (135, 46, 5216, 652, 661)), ("Fb15k237", (14541, 237, 272115, 17535, 20466)),
                      ("Wn18rr", (40943, 11, 86835, 3034, 3134))]:
        ds = getattr(datasets, cls)(batch_size=1, shuffle=False, seed=42, num_workers=0)
        assert (ds.n_entity, ds.n_relation, len(ds.train), len(ds.valid), len(ds.test)) == want


def test_train_batches_and_weights_match_reference(golden):
    from mkb_amd import datasets

    g = golden("weights.npz")
    for cls in ["Umls", "Fb15k237"]:
        ds = getattr(datasets, cls)(batch_size=256, shuffle=False, seed=42, num_workers=0)
        np.testing.assert_array_equal(ds.dataset_head.dataset.weights.numpy(), g[f"{cls}/weights"])
    ds = datasets.Umls(batch_size=256, shuffle=False, seed=42, num_workers=0)
    for i, data in enumerate(ds):
        if i == 4:
            break
        np.testing.assert_array_equal(data["sample"].numpy(), g[f"Umls/batch{i}/sample"])
        np.testing.assert_array_equal(data["weight"].numpy(), g[f"Umls/batch{i}/weight"])
        assert data["mode"] == bytes(g[f"Umls/batch{i}/mode"]).decode()
    ds = datasets.Umls(batch_size=256, shuffle=True, seed=42, num_workers=0)  # torch RandomSampler order
    for i, data in enumerate(ds):
        if i == 2:
            break
        np.testing.assert_array_equal(data["sample"].numpy(), g[f"Umls/shuffled{i}/sample"])


def test_sampler_csr_equals_oracle_dicts():
    from mkb_amd import datasets, sampling
    from oracle import sampler as osamp

    ds = datasets.Umls(batch_size=8, shuffle=False, seed=42, num_workers=0)
    th, tt = sampling.positive_triples(ds.train)
    oh, ot = osamp.positive_triples(ds.train)
    assert set(th) == set(oh) and set(tt) == set(ot)
    for k in oh:
        np.testing.assert_array_equal(th[k], oh[k])
    for k in ot:
        np.testing.assert_array_equal(tt[k], ot[k])


def test_test_dataset_items():
    """datasets/base.py:196-251: candidate ids / filter bias of TestDataset and TestDatasetRelation."""
    from mkb_amd.datasets import TestDataset, TestDatasetRelation

    ents, rels = {i: i for i in range(5)}, {0: 0, 1: 1}
    true = [(0, 0, 1), (2, 0, 1), (0, 0, 3), (0, 1, 1)]
    s, n, b, mode = TestDataset([(0, 0, 1)], true, ents, rels, "head-batch")[0]
    assert s.tolist() == [0, 0, 1] and mode == "head-batch"
    assert n.tolist() == [0, 1, 0, 3, 4] and b.tolist() == [0, 0, -100000.0, 0, 0]   # (2,0,1) is another true triple
    s, n, b, _ = TestDataset([(0, 0, 1)], true, ents, rels, "tail-batch")[0]
    assert n.tolist() == [0, 1, 2, 1, 4] and b.tolist() == [0, 0, 0, -100000.0, 0]   # (0,0,3)
    s, n, b, mode = TestDatasetRelation([(0, 0, 1)], true, ents, rels)[0]
    assert mode == "relation-batch" and n.tolist() == [[0, 0, 1], [0, 0, 1]] and b.tolist() == [0, -1]


def test_in_process_train_loaders_yield_the_worker_process_order():
    """datasets.Dataset builds its two training DataLoaders without worker processes (a batch is one indexed read:
    TrainDataset.__getitems__) and still yields the batches of the reference's num_workers=1 loaders, in their order: a
    worker-process loader draws base seed AND shuffle seed when its iterator is created, Dataset.__iter__ reproduces that."""
    import itertools

    import torch
    from torch.utils import data

    from mkb_amd import datasets
    from mkb_amd.datasets.base import TrainDataset

    def ours():
        ds = datasets.CountriesS1(batch_size=20, seed=42)
        return [(b["mode"], b["sample"].clone(), b["weight"].clone()) for _ in range(2) for b in ds]

    def workers():
        ds = datasets.CountriesS1(batch_size=20, seed=42)
        views = {m: ds._loaders[m].dataset for m in ("head-batch", "tail-batch")}
        for v in views.values():
            v.__class__ = type("PerItem", (TrainDataset,), {"__getitems__": None})  # the reference's per-triple __getitem__ path
        loaders = [data.DataLoader(views[m], batch_size=20, shuffle=True, num_workers=1, collate_fn=TrainDataset.collate_fn)
                   for m in ("head-batch", "tail-batch")]
        return [(b["mode"], b["sample"].clone(), b["weight"].clone()) for _ in range(2)
                for b in itertools.chain.from_iterable(zip(*loaders))]

    a, b = ours(), workers()
    assert len(a) == len(b) > 0
    for (ma, sa, wa), (mb, sb, wb) in zip(a, b):
        assert ma == mb and torch.equal(sa, sb) and torch.equal(wa, wb)


def test_row_lazy_replay_loop_has_no_vector_memory_load(tmp_path):
    """Round 5: the per-step constants of the row-lazy Adam replay were ordinary global reads -- the compiler issued a vector load
    and a full ``s_waitcnt vmcnt(0)`` per replayed step (two L2 round trips in every link of a serial chain; 5 us of the headline
    step).  They go through the constant address space now (``s_load``).  This compiles adam.hip for gfx950 (device side only, a
    few seconds, no GPU) and looks at the ISA: the replay loops -- the loops of ``adam_rows_catchup_kernel`` with ``v_sqrt_f32``
    and no store -- must not hold a vector-memory load, and must read their constants with scalar loads."""
    import importlib.util
    import pathlib
    import subprocess

    from mkb_amd.csrc import build as hb

    root = pathlib.Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("isa_chains", root / "tools" / "isa_chains.py")
    ic = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ic)
    asm = tmp_path / "adam.s"
    subprocess.run([hb.HIPCC, *hb.FLAGS, "-I", str(root / "include"), "--cuda-device-only", "-S",
                    str(root / "mkb_amd" / "csrc" / "adam.hip"), "-o", str(asm)], check=True, capture_output=True)
    seen = 0
    for name, body in ic.kernels(asm).items():
        if "adam_rows_catchup_kernel" not in name:
            continue
        replay = [ls for ls in ic.loops(body).values()
                  if any("v_sqrt_f32" in l for l in ls) and not any("global_store" in l or "flat_store" in l for l in ls)]
        assert replay, f"no replay loop found in {name}"
        for ls in replay:
            assert not [l for l in ls if ic.VMEM_LOAD.search(l)], f"vector load inside a replay loop of {name}"
        assert any("s_load_dwordx2" in l for ls in replay for l in ls), f"{name}: the constants do not come through the scalar cache"
        seen += 1
    assert seen == 2  # the unroll-1 and the unroll-4 instantiation


def test_weak_id_table_is_keyed_by_identity_and_forgets_dead_parameters():
    """mkb_amd._links.WeakIdTable (the side tables that tie a parameter to its row-lazy optimizer): identity keys -- two equal
    tensors are two keys --, values replaced in place, entries gone with their parameter."""
    import gc

    from mkb_amd._links import WeakIdTable

    t = WeakIdTable()
    a, b = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    t[a] = 1
    t[b] = 2
    t[a] = 3
    assert t.get(a) == 3 and t[b] == 2 and a in t and len(t) == 2
    assert t.pop(a) == 3 and a not in t and t.get(a, 7) == 7 and t.pop(a, "gone") == "gone"
    with pytest.raises(KeyError):
        t[a]
    del b
    gc.collect()
    assert len(t) == 0


def test_pipeline_host_stager_is_a_plain_copy_off_the_gpu():
    """compose.pipeline._HostStager stages host batches through page-locked buffers for a ROCm device; for any other device (the
    reference's default device="cpu") it must be the plain ``.to(device)`` of the reference's loop."""
    from mkb_amd.compose.pipeline import _HostStager

    s = _HostStager("cpu")
    x, w = torch.arange(6).reshape(2, 3), torch.ones(2)
    y, v = s(x, w)
    assert torch.equal(x, y) and torch.equal(w, v) and y.device.type == "cpu"

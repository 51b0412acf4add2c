"""Generate golden fixtures from the LIVE reference (``/root/reference``, read-only) in the build
container.  Outputs small ``.npz`` / ``.json`` files under ``tests/golden/`` -- DATA ONLY (inputs and
the reference's outputs); no reference source is copied.  The reference never travels to the GPU box;
these fixtures and the ``oracle/`` restatement (validated against them) do.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

The third-party ``river`` package (only ``river.stats.Mean`` / ``RollingMean`` are used by the reference,
``compose/pipeline.py:3``, ``evaluation/evaluation.py:5``) is absent offline; a minimal in-memory
stand-in with the same ``update`` / ``get`` behaviour is injected before importing ``mkb``.
"""
import collections
import json
import pathlib
import sys
import types

import numpy as np
import torch

OUT = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden"


def _install_river_stub():
    river = types.ModuleType("river")
    stats = types.ModuleType("river.stats")

    class Mean:
        def __init__(self):
            self.n, self.mean = 0, 0.0

        def update(self, x, w=1.0):
            self.n += w
            self.mean += (w / self.n) * (x - self.mean)  # river's running update
            return self

        def get(self):
            return self.mean

    class RollingMean:
        def __init__(self, window_size):
            self.w = collections.deque(maxlen=window_size)

        def update(self, x):
            self.w.append(x)
            return self

        def get(self):
            return sum(self.w) / len(self.w) if self.w else 0.0

    stats.Mean, stats.RollingMean = Mean, RollingMean
    river.stats = stats
    sys.modules["river"] = river
    sys.modules["river.stats"] = stats


_install_river_stub()
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import mkb  # noqa: E402
from mkb import compose, datasets, evaluation, losses, models, sampling  # noqa: E402

MODELS = ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"]
MODES = [None, "head-batch", "tail-batch"]


def npy(t):
    return None if t is None else t.detach().cpu().numpy().copy()


def gen_models():
    """5 models x 3 modes, tiny shapes, arbitrary (non-pool) negatives with duplicates;
    scores, loss, dense grads, 3-D samples, and a 3-step Adam trajectory."""
    out = {}
    N, R, hid, B, K = 40, 6, 8, 5, 7
    ents = {f"e{i}": i for i in range(N)}
    rels = {f"r{i}": i for i in range(R)}
    rs = np.random.RandomState(123)
    sample = np.stack([rs.randint(N, size=B), rs.randint(R, size=B), rs.randint(N, size=B)], 1)
    sample[3] = sample[1]  # duplicated row -> gradients must accumulate
    neg = rs.randint(N, size=(B, K))
    neg[0, 3] = neg[0, 1]
    neg[2, 0] = sample[2, 0]
    weight = rs.rand(B).astype(np.float32) + 0.1
    sample3d = np.stack([rs.randint(N, size=(B, 4)), rs.randint(R, size=(B, 4)), rs.randint(N, size=(B, 4))], 2)
    out["sample"], out["neg"], out["weight"], out["sample3d"] = sample, neg, weight, sample3d
    out["meta"] = np.array([N, R, hid, B, K])
    gamma, alpha = 3.0, 0.7
    out["gamma"], out["alpha"] = np.float32(gamma), np.float32(alpha)
    s_t, n_t, w_t = torch.LongTensor(sample), torch.LongTensor(neg), torch.tensor(weight)
    for mi, name in enumerate(MODELS):
        torch.manual_seed(1000 + mi)
        m = getattr(models, name)(hidden_dim=hid, entities=ents, relations=rels, gamma=gamma)
        if name == "pRotatE":
            with torch.no_grad():
                m.modulus.fill_(0.8)
        if name == "RotatE":  # exercise the |z| = 0 sub-gradient (torch norm backward gives 0)
            with torch.no_grad():
                m.relation_embedding[sample[4, 1]].zero_()
                m.entity_embedding[neg[4, 2]] = m.entity_embedding[sample[4, 0]]
        out[f"{name}/ent"], out[f"{name}/rel"] = npy(m.entity_embedding), npy(m.relation_embedding)
        if hasattr(m, "modulus"):
            out[f"{name}/modulus"] = npy(m.modulus)
        out[f"{name}/embedding_range"] = npy(m.embedding_range)
        out[f"{name}/score3d"] = npy(m(torch.LongTensor(sample3d)))
        for mode in MODES:
            tag = f"{name}/{mode}"
            out[f"{tag}/score"] = npy(m(s_t, None if mode is None else n_t, mode))
        for mode in MODES[1:]:
            tag = f"{name}/{mode}"
            m.zero_grad()
            pos = m(s_t)
            ng = m(s_t, n_t, mode)
            err = losses.Adversarial(alpha=alpha)(pos, ng, w_t)
            err.backward()
            out[f"{tag}/pos"], out[f"{tag}/negscore"], out[f"{tag}/loss"] = npy(pos), npy(ng), npy(err)
            out[f"{tag}/g_ent"] = npy(m.entity_embedding.grad)
            out[f"{tag}/g_rel"] = npy(m.relation_embedding.grad)
            if name == "pRotatE":
                out[f"{tag}/g_modulus"] = npy(m.modulus.grad)
            if name == "RotatE":
                assert m.modulus.grad is None
        # 3-step Adam trajectory (pipeline.py:211-240 order: step then zero_grad), alternating modes
        m.zero_grad()
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=0.01)
        lossf = losses.Adversarial(alpha=alpha)
        traj = []
        for step in range(3):
            mode = MODES[1 + step % 2]
            err = lossf(m(s_t), m(s_t, n_t, mode), w_t)
            err.backward()
            opt.step()
            opt.zero_grad()
            traj.append(err.item())
        out[f"{name}/adam/loss"] = np.array(traj, dtype=np.float32)
        out[f"{name}/adam/ent"], out[f"{name}/adam/rel"] = npy(m.entity_embedding), npy(m.relation_embedding)
        if name == "pRotatE":
            out[f"{name}/adam/modulus"] = npy(m.modulus)
    np.savez_compressed(OUT / "models.npz", **out)


def gen_init():
    """Init doctests (models/*.py): CountriesS1, hidden=3, gamma=1, torch.manual_seed(42)."""
    ds = datasets.CountriesS1(batch_size=2, seed=42)
    out = {}
    for name in MODELS:
        torch.manual_seed(42)
        m = getattr(models, name)(hidden_dim=3, entities=ds.entities, relations=ds.relations, gamma=1)
        out[f"{name}/oceania"] = npy(m.embeddings["entities"]["oceania"])
        out[f"{name}/locatedin"] = npy(m.embeddings["relations"]["locatedin"])
        out[f"{name}/repr"] = np.frombuffer(repr(m).encode(), dtype=np.uint8)
        out[f"{name}/ent"], out[f"{name}/rel"] = npy(m.entity_embedding), npy(m.relation_embedding)
    np.savez_compressed(OUT / "init.npz", **out)


def gen_sampler():
    out = {"numpy_version": np.frombuffer(np.__version__.encode(), dtype=np.uint8)}
    # raw stream known answers
    rs = np.random.RandomState(42)
    out["kat/randint4_a"] = rs.randint(4, size=10)
    out["kat/randint4_b"] = rs.randint(4, size=10)
    out["kat/randint14541"] = np.random.RandomState(42).randint(14541, size=2000)
    out["kat/randint135_seed7"] = np.random.RandomState(7).randint(135, size=1500)
    # toy doctest (negative_sampling.py:36-126)
    ents = {f"e_{i}": i for i in range(4)}
    rels = {f"r_{i}": i for i in range(4)}
    train = [(0, 0, 1), (1, 0, 2), (2, 0, 3), (3, 0, 1)]
    ns = sampling.NegativeSampling(size=5, train_triples=train, entities=ents, relations=rels, seed=42)
    smp = torch.LongTensor([[0, 0, 1], [1, 0, 2]])
    out["toy/tail"] = npy(ns.generate(smp, mode="tail-batch"))
    out["toy/head"] = npy(ns.generate(smp, mode="head-batch"))
    torch.manual_seed(42)
    m = models.RotatE(entities=ents, relations=rels, hidden_dim=3, gamma=3)
    out["toy/ent"], out["toy/rel"] = npy(m.entity_embedding), npy(m.relation_embedding)
    out["toy/score_tail"] = npy(m(smp, torch.LongTensor(out["toy/tail"]), mode="tail-batch"))
    out["toy/score_head"] = npy(m(smp, torch.LongTensor(out["toy/head"]), mode="head-batch"))
    # real graphs: (dataset, K, B, calls)
    for cls, K, B, calls in [("Umls", 16, 256, 4), ("Wn18rr", 128, 192, 4), ("Fb15k237", 256, 256, 6)]:
        ds = getattr(datasets, cls)(batch_size=B, shuffle=False, seed=42, num_workers=0)
        train = np.asarray(ds.train)
        ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities,
                                       relations=ds.relations, seed=42)
        pick = np.random.RandomState(7)
        for c in range(calls):
            idx = pick.randint(len(train), size=B)
            mode = "head-batch" if c % 2 == 0 else "tail-batch"
            neg = npy(ns.generate(torch.LongTensor(train[idx]), mode=mode))
            out[f"{cls}/{c}/idx"] = idx.astype(np.int32)
            out[f"{cls}/{c}/neg"] = neg.astype(np.int32)
        out[f"{cls}/K"] = np.array(K)
    np.savez_compressed(OUT / "sampler.npz", **out)


def gen_weights():
    out = {}
    for cls in ["Umls", "Fb15k237"]:
        ds = getattr(datasets, cls)(batch_size=256, shuffle=False, seed=42, num_workers=0)
        w = ds.dataset_head.dataset.weights
        out[f"{cls}/weights"] = torch.cat([w[i] for i in range(len(w))]).numpy()
    ds = datasets.Umls(batch_size=256, shuffle=False, seed=42, num_workers=0)
    for i, data in enumerate(ds):
        if i == 4:
            break
        out[f"Umls/batch{i}/sample"] = npy(data["sample"])
        out[f"Umls/batch{i}/weight"] = npy(data["weight"])
        out[f"Umls/batch{i}/mode"] = np.frombuffer(data["mode"].encode(), dtype=np.uint8)
    # shuffled order pin (torch RandomSampler under torch.manual_seed(seed), dataset.py:185-186)
    ds = datasets.Umls(batch_size=256, shuffle=True, seed=42, num_workers=0)
    for i, data in enumerate(ds):
        if i == 2:
            break
        out[f"Umls/shuffled{i}/sample"] = npy(data["sample"])
    np.savez_compressed(OUT / "weights.npz", **out)


def gen_pipeline():
    """Pipeline.learn on CountriesS1 (setup of compose/pipeline.py:79-129), every step recorded."""
    torch.manual_seed(42)
    ds = datasets.CountriesS1(batch_size=20, seed=42)
    model = models.RotatE(hidden_dim=5, entities=ds.entities, relations=ds.relations, gamma=3)
    out = {"ent0": npy(model.entity_embedding), "rel0": npy(model.relation_embedding)}
    ns = sampling.NegativeSampling(size=4, train_triples=ds.train, entities=ds.entities,
                                   relations=ds.relations, seed=42)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=0.00005)
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations,
                               batch_size=8)
    lossf = losses.Adversarial(alpha=0.5)
    rec = {"sample": [], "neg": [], "mode": [], "loss": []}
    gen0 = ns.generate

    def gen(sample, mode):
        n = gen0(sample=sample, mode=mode)
        rec["sample"].append(npy(sample)); rec["neg"].append(npy(n)); rec["mode"].append(mode == "head-batch")
        return n

    ns.generate = gen

    class L:
        def __call__(self, p, n, w):
            e = lossf(p, n, w)
            rec["loss"].append(e.item())
            return e

    pipe = compose.Pipeline(epochs=3, eval_every=1, early_stopping_rounds=3)
    pipe = pipe.learn(model=model, dataset=ds, evaluation=ev, sampling=ns, optimizer=opt, loss=L())
    nfull = [i for i, s in enumerate(rec["sample"]) if s.shape[0] == 20]
    out["steps"] = np.array(len(rec["sample"]))
    for i, (s, n) in enumerate(zip(rec["sample"], rec["neg"])):
        out[f"step{i}/sample"], out[f"step{i}/neg"] = s.astype(np.int32), n.astype(np.int32)
    out["mode_head"] = np.array(rec["mode"])
    out["loss"] = np.array(rec["loss"], dtype=np.float64)
    out["ent_final"], out["rel_final"] = npy(model.entity_embedding), npy(model.relation_embedding)
    np.savez_compressed(OUT / "pipeline_countries.npz", **out)
    with open(OUT / "pipeline_countries.json", "w") as f:
        json.dump({"valid_scores": pipe.valid_scores, "test_scores": pipe.test_scores,
                   "torch": torch.__version__, "numpy": np.__version__}, f, indent=1)


def gen_eval():
    """evaluation.py:43-119 known-answer (5 epochs, no zero_grad, Adam lr .5) + score rows."""
    torch.manual_seed(42)
    train = [(0, 0, 1), (0, 1, 1), (2, 0, 3), (2, 1, 3)]
    valid = [(0, 0, 1), (2, 1, 3)]
    test = [(0, 0, 1), (2, 1, 3)]
    ents = {f"e{i}": i for i in range(4)}
    rels = {"r0": 0, "r1": 1}
    ds = datasets.Dataset(train=train, valid=valid, test=test, entities=ents, relations=rels, batch_size=2,
                          seed=42, shuffle=False)
    ns = sampling.NegativeSampling(size=2, train_triples=ds.train, entities=ds.entities,
                                   relations=ds.relations, seed=42)
    model = models.RotatE(hidden_dim=3, entities=ds.entities, relations=ds.relations, gamma=1)
    out = {"ent0": npy(model.entity_embedding), "rel0": npy(model.relation_embedding)}
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=0.5)
    lossf = losses.Adversarial(alpha=0.5)
    for _ in range(5):
        for data in ds:
            s, w, mode = data["sample"], data["weight"], data["mode"]
            lossf(model(s), model(s, ns.generate(sample=s, mode=mode), mode), w).backward()
            opt.step()
    model = model.eval()
    out["ent"], out["rel"] = npy(model.entity_embedding), npy(model.relation_embedding)
    ev = evaluation.Evaluation(true_triples=train + valid + test, entities=ents, relations=rels, batch_size=2)
    res = {"toy_eval": ev.eval(model=model, dataset=test),
           "toy_eval_relations": ev.eval_relations(model=model, dataset=test)}
    # a larger filtered-ranking case: CountriesS1 with a seeded, untrained ComplEx and TransE
    cds = datasets.CountriesS1(batch_size=20, seed=42)
    for name in ["TransE", "ComplEx", "RotatE"]:
        torch.manual_seed(7)
        m = getattr(models, name)(hidden_dim=6, entities=cds.entities, relations=cds.relations, gamma=4).eval()
        cev = evaluation.Evaluation(true_triples=cds.true_triples, entities=cds.entities,
                                    relations=cds.relations, batch_size=8)
        res[f"countries/{name}/test"] = cev.eval(model=m, dataset=cds.test)
        res[f"countries/{name}/valid"] = cev.eval(model=m, dataset=cds.valid)
        out[f"countries/{name}/ent"], out[f"countries/{name}/rel"] = npy(m.entity_embedding), npy(m.relation_embedding)
    np.savez_compressed(OUT / "evaluation.npz", **out)
    with open(OUT / "evaluation.json", "w") as f:
        json.dump(res, f, indent=1)


def headline_tables(n_entity=14541, n_relation=237, hidden=1000, gamma=9.0, seed=2024):
    """Tables of the headline-shape slice fixture, drawn with numpy's legacy generator (bit-stable across numpy
    versions and platforms) instead of torch's CPU RNG stream, so the test that re-creates them never depends on the
    torch build: U(-(gamma + 2) / hidden, +(gamma + 2) / hidden) like models/base.py:81-100."""
    rs = np.random.RandomState(seed)
    r = (gamma + 2.0) / hidden
    ent = rs.uniform(-r, r, size=(n_entity, 2 * hidden)).astype(np.float32)
    rel = rs.uniform(-r, r, size=(n_relation, hidden)).astype(np.float32)
    return ent, rel


def gen_headline_slice():
    """One real-shape RotatE batch row-slice on FB15k-237 (hidden=1000, K=256): scores, loss AND gradients of a 16-row
    slice from the live reference (the full 1024-row reference step takes ~80 s on CPU; the slice keeps the fixture
    small).  The dense entity gradient (116 MB) is stored as: ids of its non-zero rows, every 8th column of those rows,
    and their float64 row sums / squared norms; the relation gradient's non-zero rows are stored whole."""
    ds = datasets.Fb15k237(batch_size=16, shuffle=False, seed=42, num_workers=0)
    m = models.RotatE(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9)
    ent, rel = headline_tables()
    with torch.no_grad():
        m.entity_embedding.copy_(torch.from_numpy(ent))
        m.relation_embedding.copy_(torch.from_numpy(rel))
    ns = sampling.NegativeSampling(size=256, train_triples=ds.train, entities=ds.entities,
                                   relations=ds.relations, seed=42)
    train = np.asarray(ds.train)
    rs = np.random.RandomState(7)
    idx = rs.randint(len(train), size=16)
    weight = (rs.rand(16) + 0.1).astype(np.float32)
    s = torch.LongTensor(train[idx])
    out = {"idx": idx.astype(np.int32), "weight": weight, "alpha": np.float32(1.0),
           "table_seed": np.int32(2024)}
    for mode in ["head-batch", "tail-batch"]:
        neg = ns.generate(s, mode=mode)
        m.zero_grad()
        pos_score, neg_score = m(s), m(s, neg, mode)
        err = losses.Adversarial(alpha=1.0)(pos_score, neg_score, torch.from_numpy(weight))
        err.backward()
        out[f"{mode}/neg"] = npy(neg).astype(np.int32)
        out[f"{mode}/score"] = npy(neg_score)
        out[f"{mode}/pos"] = npy(pos_score)
        out[f"{mode}/loss"] = npy(err)
        ge, gr = npy(m.entity_embedding.grad), npy(m.relation_embedding.grad)
        rows = np.flatnonzero(np.abs(ge).sum(1) > 0)
        out[f"{mode}/g_ent_rows"] = rows.astype(np.int32)
        out[f"{mode}/g_ent_cols8"] = ge[rows][:, ::8].copy()
        out[f"{mode}/g_ent_rowsum"] = ge[rows].astype(np.float64).sum(1)
        out[f"{mode}/g_ent_rowsq"] = (ge[rows].astype(np.float64) ** 2).sum(1)
        rrows = np.flatnonzero(np.abs(gr).sum(1) > 0)
        out[f"{mode}/g_rel_rows"] = rrows.astype(np.int32)
        out[f"{mode}/g_rel"] = gr[rrows].copy()
    np.savez_compressed(OUT / "headline_slice.npz", **out)


def gen_eval_headline():
    """Filtered ranking at the headline size from the LIVE reference: FB15k-237, hidden 1000, all five models, 8 test triples x both
    sides, each against all 14,541 entities -- ``datasets.base.TestDataset`` items (base.py:196-241) collated and ranked by the
    arithmetic of ``Evaluation.compute_score`` (evaluation.py:232-263).  Stored per (model, mode): the ranks, the biased scores of
    every 16th column, the unbiased score of the target; tables come from tests/util_gpu_tables.py::eval_tables (numpy legacy
    generator), so the fixture holds no table."""
    sys.path.insert(0, str(OUT.parent))
    from util_gpu_tables import eval_tables
    from mkb.datasets import base as ref_base

    ds = datasets.Fb15k237(batch_size=8, shuffle=False, seed=42)
    test = np.asarray(ds.test, dtype=np.int64)
    idx = np.random.RandomState(11).choice(len(test), size=8, replace=False)
    out = {"idx": idx.astype(np.int32), "table_seed": np.int32(77), "gamma": np.float32(9.0), "hidden": np.int32(1000),
           "column_stride": np.int32(16)}
    triples = [tuple(int(v) for v in test[i]) for i in idx]
    for name in MODELS:
        ent, rel, modulus = eval_tables(name, seed=77)
        m = getattr(models, name)(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9).eval()
        with torch.no_grad():
            m.entity_embedding.copy_(torch.from_numpy(ent))
            m.relation_embedding.copy_(torch.from_numpy(rel))
            if modulus is not None:
                assert np.array_equal(npy(m.modulus), modulus), (npy(m.modulus), modulus)
        for mode in ["head-batch", "tail-batch"]:
            td = ref_base.TestDataset(triples=triples, true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations,
                                      mode=mode)
            data = ref_base.TestDataset.collate_fn([td[i] for i in range(len(triples))])
            with torch.no_grad():
                score = m(sample=data["sample"], negative_sample=data["negative_sample"], mode=mode)
                target = data["sample"][:, 0] if mode == "head-batch" else data["sample"][:, 2]
                out[f"{name}/{mode}/target_score"] = npy(score[torch.arange(len(triples)), target])
                score += data["filter_bias"]
                argsort = torch.argsort(score, dim=1, descending=True)
                ranks = [1 + (argsort[i, :] == target[i]).nonzero().item() for i in range(len(triples))]
            out[f"{name}/{mode}/ranks"] = np.asarray(ranks, dtype=np.int64)
            out[f"{name}/{mode}/biased_scores_every_16th"] = npy(score[:, ::16])
            out[f"{name}/{mode}/n_filtered"] = npy((data["filter_bias"] != 0).sum(dim=1)).astype(np.int32)
            print(name, mode, ranks, file=sys.stderr)
    np.savez_compressed(OUT / "eval_headline.npz", **out)


def gen_distill():
    """Distillation (distillation/distillation.py:438-683, kdmkb_model.py:286-360) captured from the live reference:
    (a) KlDivergence values + student gradients on random score matrices; (b) Distillation.distill with UniformSampling on
    Umls RotatE hidden 3 (the reference doctest: loss 1.3066): sampled candidate tensors, loss, student gradients;
    (c) two partially overlapping toy graphs (only some entities / relations shared); (d) KdmkbModel.forward, 3 steps on two
    CountriesS1 copies with UniformSampling swapped in for the faiss sampler: per-step losses and final tables."""
    from mkb import distillation
    out, js = {}, {}
    g = torch.Generator().manual_seed(0)
    for tag, (n, m, T) in {"a": (5, 7, 1.0), "b": (3, 130, 2.5), "c": (1, 2, 0.5)}.items():
        s = (torch.randn(n, m, generator=g) * 3).requires_grad_(True)
        t = torch.randn(n, m, generator=g) * 3
        loss = losses.KlDivergence()(student_score=s, teacher_score=t, T=T)
        loss.backward()
        out[f"kl/{tag}/student"], out[f"kl/{tag}/teacher"], out[f"kl/{tag}/T"] = npy(s), npy(t), np.float32(T)
        out[f"kl/{tag}/loss"], out[f"kl/{tag}/dstudent"] = npy(loss), npy(s.grad)

    # (b) the reference doctest setup
    torch.manual_seed(42)
    ds = datasets.Umls(batch_size=3, shuffle=False, seed=42)
    teacher = models.RotatE(hidden_dim=3, entities=ds.entities, relations=ds.relations, gamma=6)
    student = models.RotatE(hidden_dim=3, entities=ds.entities, relations=ds.relations, gamma=6)
    proc = distillation.Distillation(teacher_entities=ds.entities, student_entities=ds.entities, teacher_relations=ds.relations,
                                     student_relations=ds.relations,
                                     sampling=distillation.UniformSampling(batch_size_entity=3, batch_size_relation=3, seed=42))
    data = next(iter(ds))
    loss = proc.distill(teacher=teacher, student=student, sample=data["sample"])
    loss.backward()
    out["umls/teacher_ent"], out["umls/teacher_rel"] = npy(teacher.entity_embedding), npy(teacher.relation_embedding)
    out["umls/student_ent"], out["umls/student_rel"] = npy(student.entity_embedding), npy(student.relation_embedding)
    out["umls/sample"], out["umls/loss"] = npy(data["sample"]), npy(loss)
    out["umls/g_ent"], out["umls/g_rel"] = npy(student.entity_embedding.grad), npy(student.relation_embedding.grad)
    js["umls_doctest_loss"] = round(float(loss), 4)

    # (c) partial overlap: teacher graph e0..e5 / r0..r2, student graph e3..e8 / r1..r3 (different ids for shared labels)
    t_ents = {f"e{i}": i for i in range(6)}
    s_ents = {f"e{i}": j for j, i in enumerate([7, 3, 8, 5, 4, 6])}
    t_rels = {f"r{i}": i for i in range(3)}
    s_rels = {"r3": 0, "r1": 1, "r2": 2}
    torch.manual_seed(7)
    teacher = models.TransE(hidden_dim=4, entities=t_ents, relations=t_rels, gamma=3)
    student = models.DistMult(hidden_dim=5, entities=s_ents, relations=s_rels, gamma=3)
    proc = distillation.Distillation(teacher_entities=t_ents, student_entities=s_ents, teacher_relations=t_rels,
                                     student_relations=s_rels,
                                     sampling=distillation.UniformSampling(batch_size_entity=2, batch_size_relation=2, seed=5))
    sample = torch.tensor([[3, 1, 4], [0, 1, 3], [5, 2, 3], [4, 0, 5], [3, 2, 5]])
    loss = proc.distill(teacher=teacher, student=student, sample=sample)
    loss.backward()
    out["part/teacher_ent"], out["part/teacher_rel"] = npy(teacher.entity_embedding), npy(teacher.relation_embedding)
    out["part/student_ent"], out["part/student_rel"] = npy(student.entity_embedding), npy(student.relation_embedding)
    out["part/sample"], out["part/loss"] = npy(sample), npy(loss)
    out["part/g_ent"], out["part/g_rel"] = npy(student.entity_embedding.grad), npy(student.relation_embedding.grad)
    js["part_available"] = [proc.available(*row) for row in sample.tolist()]

    # (d) KdmkbModel.forward with the uniform sampler (the reference hard-wires the faiss one: swap the name it looks up)
    from mkb.distillation import kdmkb_model as km
    km.FastTopKSampling = distillation.UniformSampling
    torch.manual_seed(42)
    d1 = datasets.CountriesS1(batch_size=8, seed=42)
    d2 = datasets.CountriesS1(batch_size=8, seed=42)
    m1 = models.TransE(hidden_dim=6, entities=d1.entities, relations=d1.relations, gamma=3)
    m2 = models.RotatE(hidden_dim=4, entities=d2.entities, relations=d2.relations, gamma=3)
    out["kd/m1_ent"], out["kd/m1_rel"], out["kd/m2_ent"], out["kd/m2_rel"] = (npy(m1.entity_embedding), npy(m1.relation_embedding),
                                                                              npy(m2.entity_embedding), npy(m2.relation_embedding))
    mods, dsets = collections.OrderedDict(a=m1, b=m2), collections.OrderedDict(a=d1, b=d2)
    kd = km.KdmkbModel(models=mods, datasets=dsets, lr={"a": 1e-2, "b": 1e-2}, alpha_kl={"a": 0.3, "b": 0.6},
                       alpha_adv={"a": 0.5, "b": 0.5}, negative_sampling_size={"a": 4, "b": 4}, batch_size_entity={"a": 5, "b": 5},
                       batch_size_relation={"a": 2, "b": 2}, n_random_entities={"a": 3, "b": 3}, n_random_relations={"a": 1, "b": 1},
                       device="cpu", seed=42)
    steps = []
    for step in range(3):
        w = {"a": 0.3, "b": 0.6}
        before = {k: kd.metrics[k].get() for k in mods}
        kd.forward(dsets, mods, w)
        steps.append({k: kd.metrics[k].w[-1] for k in mods})
    js["kd_step_losses"] = steps
    out["kd/m1_ent_after"], out["kd/m2_ent_after"] = npy(m1.entity_embedding), npy(m2.entity_embedding)
    out["kd/m1_rel_after"], out["kd/m2_rel_after"] = npy(m1.relation_embedding), npy(m2.relation_embedding)
    np.savez_compressed(OUT / "distill.npz", **out)
    (OUT / "distill.json").write_text(json.dumps(js, indent=1))


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    which = sys.argv[1:] or ["models", "init", "sampler", "weights", "pipeline", "eval", "headline_slice", "distill", "eval_headline"]
    for w in which:
        print("generating", w, file=sys.stderr)
        globals()[f"gen_{w}"]()

"""-m gpu: general (arbitrary-candidate) scoring path, loss and Adam through the C ABI vs the oracle and the
golden vectors captured from the live reference.  Tolerance: 1e-4 absolute fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from mkb_amd import _links

pytestmark = pytest.mark.gpu

from util_gpu import grad_close  # noqa: E402  (tests/ is on sys.path: conftest.py)

MODELS = ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"]
MODES = [None, "head-batch", "tail-batch"]
ATOL = 1e-4


def _golden_model(g, name):
    from util_gpu import make_model

    N, R, hid, B, K = (int(v) for v in g["meta"])
    mod = g[f"{name}/modulus"] if f"{name}/modulus" in g.files else None
    return make_model(name, g[f"{name}/ent"], g[f"{name}/rel"], hid, float(g["gamma"]), mod)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("mode", MODES)
def test_scores_vs_reference_golden(golden, name, mode):
    g = golden("models.npz")
    m = _golden_model(g, name)
    s = torch.as_tensor(g["sample"]).cuda()
    n = torch.as_tensor(g["neg"]).cuda()
    got = m(s, None if mode is None else n, mode)
    np.testing.assert_allclose(got.detach().cpu().numpy(), g[f"{name}/{mode}/score"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("name", MODELS)
def test_3d_sample(golden, name):
    g = golden("models.npz")
    m = _golden_model(g, name)
    got = m(torch.as_tensor(g["sample3d"]).cuda())
    np.testing.assert_allclose(got.detach().cpu().numpy(), g[f"{name}/score3d"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("mode", MODES[1:])
def test_loss_and_dense_grads_vs_reference_golden(golden, name, mode):
    from mkb_amd import losses

    g = golden("models.npz")
    m = _golden_model(g, name)
    s, n = torch.as_tensor(g["sample"]).cuda(), torch.as_tensor(g["neg"]).cuda()
    w = torch.as_tensor(g["weight"]).cuda()
    err = losses.Adversarial(alpha=float(g["alpha"]))(m(s), m(s, n, mode), w)
    err.backward()
    tag = f"{name}/{mode}"
    np.testing.assert_allclose(err.item(), g[f"{tag}/loss"], rtol=0, atol=1e-5)
    grad_close(m.entity_embedding.grad.cpu().numpy(), g[f"{tag}/g_ent"])
    grad_close(m.relation_embedding.grad.cpu().numpy(), g[f"{tag}/g_rel"], rtol=1e-4)
    if name == "pRotatE":
        np.testing.assert_allclose(m.modulus.grad.cpu().numpy(), g[f"{tag}/g_modulus"], rtol=1e-4)
    if name == "RotatE":
        assert m.modulus.grad is None  # unused parameter, as in the reference


@pytest.mark.parametrize("name", MODELS)
def test_adam_trajectory_vs_reference_golden(golden, name):
    """3 x (pos fwd, neg fwd, Adversarial, backward, mkb_adam_step + zero) == reference torch.optim.Adam."""
    from mkb_amd import losses, optim

    g = golden("models.npz")
    m = _golden_model(g, name)
    s, n = torch.as_tensor(g["sample"]).cuda(), torch.as_tensor(g["neg"]).cuda()
    w = torch.as_tensor(g["weight"]).cuda()
    opt = optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=0.01)
    lossf = losses.Adversarial(alpha=float(g["alpha"]))
    traj = []
    for step in range(3):
        mode = MODES[1 + step % 2]
        err = lossf(m(s), m(s, n, mode), w)
        err.backward()
        opt.step()
        opt.zero_grad()
        traj.append(err.item())
    np.testing.assert_allclose(traj, g[f"{name}/adam/loss"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(m.entity_embedding.detach().cpu().numpy(), g[f"{name}/adam/ent"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(m.relation_embedding.detach().cpu().numpy(), g[f"{name}/adam/rel"], rtol=0, atol=1e-5)
    if name == "pRotatE":
        np.testing.assert_allclose(m.modulus.detach().cpu().numpy(), g[f"{name}/adam/modulus"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("mode", MODES[1:])
@pytest.mark.parametrize("shape", [(300, 7, 64, 33, 16), (2000, 11, 250, 64, 50)])
def test_random_problem_vs_oracle(name, mode, shape):
    """Seeded mid-size problems (odd dims, duplicates inevitable) vs the torch-fp32 oracle and the float64
    closed form."""
    from mkb_amd import losses
    from oracle import closed, scoring
    from util_gpu import make_model, oracle_tables, random_problem

    N, R, hid, B, K = shape
    ent, rel, s, n, w, mod = random_problem(name, N, R, hid, B, K, seed=MODELS.index(name) * 7 + len(mode) + N)
    m = make_model(name, ent, rel, hid, 6.0, mod)
    tb = oracle_tables(name, ent, rel, hid, 6.0, mod)
    ref = scoring.train_step_grads(tb, s, n, w, mode, 1.0, fast_norm=True)
    pos, neg = m(s.cuda()), m(s.cuda(), n.cuda(), mode)
    np.testing.assert_allclose(pos.detach().cpu().numpy(), ref["pos"].numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(neg.detach().cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
    err = losses.Adversarial(alpha=1.0)(pos, neg, w.cuda())
    err.backward()
    np.testing.assert_allclose(err.item(), ref["loss"].item(), rtol=0, atol=1e-5)
    c = closed.train_step_grads(name, ent.numpy(), rel.numpy(), s.numpy(), n.numpy(), w.numpy(), mode, 1.0, 6.0, hid,
                                None if mod is None else mod.numpy())
    ge = m.entity_embedding.grad.cpu().numpy()
    gr = m.relation_embedding.grad.cpu().numpy()
    tol = 1e-5 * max(1.0, float(np.abs(c["g_ent"]).max()))   # pRotatE / RotatE divide by range/pi ~ 1e-2
    np.testing.assert_allclose(ge, ref["g_ent"].numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(ge, c["g_ent"], rtol=0, atol=tol)
    np.testing.assert_allclose(gr, c["g_rel"], rtol=1e-4, atol=tol)


def test_cpu_tensors_are_rejected():
    from mkb_amd import models

    m = models.TransE(hidden_dim=4, entities={0: 0, 1: 1}, relations={0: 0}, gamma=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.tensor([[0, 0, 1]]))


def test_row_lazy_adam_is_bitwise_dense_adam():
    """mkb_adam_rows_* (deferred zero-gradient steps) == mkb_adam_step (dense) bit for bit on a sparse-gradient
    schedule with repeated ids, rows touched once, rows never touched, and a mid-run flush."""
    from mkb_amd import optim

    torch.manual_seed(0)
    N, D = 5000, 37
    p0 = torch.randn(N, D, device="cuda") * 0.01
    pd = torch.nn.Parameter(p0.clone())
    pl = torch.nn.Parameter(p0.clone())
    od = optim.Adam([pd], lr=3e-3)
    ol = optim.Adam([pl], lr=3e-3, lazy_rows=True)
    assert _links.owner(pl) is ol
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(40):
        ids = torch.randint(0, 600 if step % 3 else N, (257,), device="cuda", generator=g)   # duplicates inside
        ids[-1] = ids[0]
        ol.catch_up(pl, ids)                                     # what FusedTrainStep does before the forward pass
        assert torch.equal(pl.data[ids], pd.data[ids])           # rows about to be read are current
        grad = torch.zeros(N, D, device="cuda")
        grad[ids] = torch.randn(ids.numel(), D, device="cuda", generator=g)
        pd.grad, pl.grad = grad.clone(), grad.clone()
        _links.mark_touched(pl, ids)
        od.step(); ol.step()
        assert float(pl.grad.abs().sum()) == 0.0
        if step == 17:
            ol.flush()
            assert torch.equal(pl.data, pd.data)
    ol.flush()
    assert torch.equal(pl.data, pd.data)
    assert torch.equal(ol.state[pl]["m"], od.state[pd]["m"]) and torch.equal(ol.state[pl]["v"], od.state[pd]["v"])


def test_sampler_emits_touched_rows_and_adam_rider_equals_separate_launches():
    """mkb_sampler_generate's `touched` list is pool | heads | tails of the batch, and a relation table that rides
    the row-lazy Adam launch (mkb_adam_dense_t) ends bit-identical to one stepped by its own mkb_adam_step launch."""
    from mkb_amd import datasets, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    ds = datasets.Umls(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    ns = sampling.NegativeSampling(size=8, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=3)
    s = train[:64].contiguous()
    neg = ns.generate(s, "tail-batch")
    info = neg._mkb_pool
    want = torch.cat([info.pool, s[:, 0], s[:, 2]])
    assert torch.equal(info.touched, want)

    results = []
    for ride in (True, False):
        g = torch.Generator(device="cpu").manual_seed(5)
        ent = torch.nn.Parameter(torch.randn(5000, 64, generator=g).cuda())   # >= 4096 rows: steps row-lazily
        rel = torch.nn.Parameter(torch.randn(37, 64, generator=g).cuda())
        opt = optim.Adam([ent, rel], lr=1e-2, lazy_rows=True)
        if not ride:
            opt._rider = lambda: (None, None)
        for it in range(5):
            ids = torch.randperm(5000, generator=g)[:300].cuda()  # distinct: index_put with duplicates is order-dependent
            ent.grad = torch.zeros_like(ent)
            ent.grad[ids] = torch.randn(300, 64, generator=g).cuda()
            rel.grad = torch.randn(37, 64, generator=g).cuda()
            opt.catch_up(ent, ids)
            _links.mark_touched(ent, ids)
            opt.step()
            opt.zero_grad()
            assert rel.grad.abs().max().item() == 0.0  # zero_grad is fused into both routes
        opt.flush()
        torch.cuda.synchronize()
        results.append((ent.detach().cpu().numpy().copy(), rel.detach().cpu().numpy().copy()))
    assert np.array_equal(results[0][0], results[1][0])
    assert np.array_equal(results[0][1], results[1][1])
    # and both equal torch.optim.Adam on the same gradients to a few ulp of the update
    g = torch.Generator(device="cpu").manual_seed(5)
    ent = torch.nn.Parameter(torch.randn(5000, 64, generator=g).cuda())
    rel = torch.nn.Parameter(torch.randn(37, 64, generator=g).cuda())
    ref = torch.optim.Adam([ent, rel], lr=1e-2)
    for it in range(5):
        ids = torch.randperm(5000, generator=g)[:300].cuda()  # distinct: index_put with duplicates is order-dependent
        ent.grad = torch.zeros_like(ent)
        ent.grad[ids] = torch.randn(300, 64, generator=g).cuda()
        rel.grad = torch.randn(37, 64, generator=g).cuda()
        ref.step()
    np.testing.assert_allclose(results[0][0], ent.detach().cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(results[0][1], rel.detach().cpu().numpy(), rtol=0, atol=2e-6)


def test_optimizer_and_sampler_checkpoint_resume_is_exact():
    """state_dict / load_state_dict of mkb_amd.optim.Adam (row-lazy and dense parameters) + the sampler's generator state:
    a run resumed from a checkpoint continues bit-identically to the uninterrupted run."""
    from mkb_amd import datasets, optim, sampling

    ds = datasets.Umls(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    mk_s = lambda: sampling.NegativeSampling(size=24, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=9)

    def make():
        g = torch.Generator(device="cpu").manual_seed(1)
        ent = torch.nn.Parameter(torch.randn(5000, 64, generator=g).cuda())  # row-lazy
        rel = torch.nn.Parameter(torch.randn(46, 64, generator=g).cuda())    # dense
        return ent, rel, optim.Adam([ent, rel], lr=1e-3, lazy_rows=True)

    def run(ent, rel, opt, sam, steps):
        for it in steps:
            s = train[it * 61: it * 61 + 61].contiguous()
            neg = sam.generate_with_catch_up(s, "head-batch" if it % 2 else "tail-batch", opt, ent)
            ids = neg._mkb_pool.touched
            ent.grad = torch.zeros_like(ent)
            ent.grad[torch.unique(ids)] = 0.125 * (it + 1)
            rel.grad = torch.full_like(rel, 0.25 * (it + 1))
            _links.mark_touched(ent, ids)
            opt.step()
            opt.zero_grad()

    ent, rel, opt = make()
    sam = mk_s()
    run(ent, rel, opt, sam, range(4))
    ckpt = {"ent": ent.detach().clone(), "rel": rel.detach().clone(), "opt": opt.state_dict(), "rng": sam.get_state()}
    ckpt["ent"] = ent.detach().clone()  # (state_dict flushed the pending row-lazy steps into the table)
    run(ent, rel, opt, sam, range(4, 8))
    opt.flush()
    want = (ent.detach().clone(), rel.detach().clone())

    ent2, rel2, opt2 = make()
    with torch.no_grad():
        ent2.copy_(ckpt["ent"]); rel2.copy_(ckpt["rel"])
    opt2.load_state_dict(ckpt["opt"])
    sam2 = mk_s()
    sam2.set_state(*ckpt["rng"], device=torch.device("cuda", 0))
    run(ent2, rel2, opt2, sam2, range(4, 8))
    opt2.flush()
    assert torch.equal(ent2.detach(), want[0]) and torch.equal(rel2.detach(), want[1])
    ka, pa = sam.get_state(); kb, pb = sam2.get_state()
    assert pa == pb and np.array_equal(ka, kb)


def test_make_prediction_reference_doctest_known_answer():
    """utils/predict.py:69-96: TransE hidden 3 on Umls (seed 42), scores of test[:3] -> tensor([-2.4270, -2.1356, -2.4053])."""
    from mkb_amd import datasets, models, utils

    torch.manual_seed(42)
    dataset = datasets.Umls(batch_size=2)
    model = models.TransE(entities=dataset.entities, relations=dataset.relations, hidden_dim=3, gamma=6).cuda()
    got = utils.make_prediction(model=model, dataset=dataset.test[:3], batch_size=20, device="cuda")
    np.testing.assert_allclose(got.cpu().numpy(), [-2.4270, -2.1356, -2.4053], rtol=0, atol=1e-4)
    batches = list(utils.FetchToPredict(dataset=[(0, 0, 1), (1, 0, 2), (1, 1, 3)], batch_size=2))
    assert [b.tolist() for b in batches] == [[[0, 0, 1], [1, 0, 2]], [[1, 1, 3]]]  # predict.py:27-31


def test_top_k_reference_doctest_known_answers():
    """utils/top_k.py:17-57: RotatE hidden 4 gamma 3 on CountriesS1 (seed 42): the doctest's top heads / relations / tails."""
    from mkb_amd import datasets, models, utils

    torch.manual_seed(42)
    dataset = datasets.CountriesS1(batch_size=2, seed=42)
    model = models.RotatE(entities=dataset.entities, relations=dataset.relations, gamma=3, hidden_dim=4).cuda()
    top_k = utils.TopK(entities=dataset.entities, relations=dataset.relations, device="cuda")
    assert top_k.top_heads(k=4, model=model, relation="neighbor", tail="western_africa") == [
        "mauritius", "são_tomé_and_príncipe", "guinea-bissau", "saint_kitts_and_nevis"]
    assert top_k.top_relations(k=4, model=model, head="azerbaijan", tail="western_africa") == ["locatedin", "neighbor"]
    assert top_k.top_tails(k=4, model=model, head="western_africa", relation="neighbor") == [
        "afghanistan", "barbados", "taiwan", "new_caledonia"]


def test_model_save_after_row_lazy_training_holds_only_the_model(tmp_path):
    """models/base.py:41-46 (``save`` pickles the model on the CPU).  The row-lazy optimizer used to hang off the entity
    Parameter, so the pickle swallowed its moments / replay constants and -- with a sampler drawn ahead -- died on the
    ctypes handle.  The links now live in mkb_amd/_links.py: the file holds the model alone and loads without a GPU."""
    import pickle

    from mkb_amd import compose, datasets, evaluation, losses, models, optim, sampling

    ds = datasets.CountriesS1(batch_size=64, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(0)
    m = models.TransE(hidden_dim=32, entities=ds.entities, relations=ds.relations, gamma=6.0).cuda()
    big = torch.nn.Parameter(torch.zeros(5000, 32, device="cuda"))  # the 271-row table stays dense: add a row-lazy one
    opt = optim.Adam([p for p in m.parameters() if p.requires_grad] + [big], lr=1e-3, lazy_rows=True)
    assert _links.owner(big) is opt and not hasattr(big, "_mkb_lazy")
    ns = sampling.NegativeSampling(size=8, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    compose.Pipeline(epochs=1, eval_every=10).learn(model=m, dataset=ds, sampling=ns, optimizer=opt, loss=losses.Adversarial(alpha=0.5),
                                                    evaluation=evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities,
                                                                                     relations=ds.relations, batch_size=64, device="cuda"))
    assert opt.draw_ahead is None or opt.draw_ahead is ns
    path = tmp_path / "model.pkl"
    want = m.entity_embedding.detach().cpu().clone()
    m.save(path)
    assert path.stat().st_size < 4 * (271 * 32 + 2 * 32) * 4 + 200_000  # the two tables and the label dicts, not an optimizer
    loaded = pickle.loads(path.read_bytes())
    assert type(loaded).__name__ == "TransE" and not loaded.entity_embedding.is_cuda
    assert torch.equal(loaded.entity_embedding.detach(), want)
    assert not any(k.startswith("_mkb") for k in vars(loaded.entity_embedding))


def test_row_lazy_model_save_with_a_sampler_drawn_ahead(tmp_path):
    """The documented fast path: lazy_rows Adam + draw_ahead sampler on a big table, then save()."""
    import pickle

    from mkb_amd import models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    N, R = 6000, 5
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    rs = np.random.RandomState(0)
    train = np.stack([rs.randint(N, size=4000), rs.randint(R, size=4000), rs.randint(N, size=4000)], 1)
    torch.manual_seed(0)
    m = models.RotatE(hidden_dim=16, entities=ents, relations=rels, gamma=6.0).cuda()
    ns = sampling.NegativeSampling(size=16, train_triples=train, entities=ents, relations=rels, seed=1)
    opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-3, lazy_rows=True, draw_ahead=ns)
    step = FusedTrainStep(m, 1.0)
    t = torch.as_tensor(train).cuda()
    for i in range(3):
        step.sampled(t[i * 64: (i + 1) * 64].contiguous(), torch.ones(64, device="cuda"), ns, "head-batch" if i % 2 else "tail-batch")
        opt.step()
        opt.zero_grad()
    dense = m.entity_embedding.detach().clone()
    opt.flush()
    path = tmp_path / "m.pkl"
    m.save(path)  # used to raise: ctypes objects containing pointers cannot be pickled
    loaded = pickle.loads(path.read_bytes())
    assert path.stat().st_size < 2 * N * 32 * 4
    np.testing.assert_array_equal(loaded.entity_embedding.detach().numpy(), m.entity_embedding.detach().cpu().numpy())
    assert dense.shape == loaded.entity_embedding.shape


def test_two_fused_steps_before_one_optimizer_step_accumulate_their_touched_rows():
    """Gradient accumulation with the row-lazy optimizer: the rows of BOTH batches take the step and are cleared
    (the touched list of the first batch used to be overwritten by the second)."""
    from mkb_amd import models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    N, R = 5000, 4
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    rs = np.random.RandomState(1)
    train = np.stack([rs.randint(N, size=3000), rs.randint(R, size=3000), rs.randint(N, size=3000)], 1)
    t = torch.as_tensor(train).cuda()
    w = torch.ones(32, device="cuda")

    def run(lazy):
        torch.manual_seed(3)
        m = models.TransE(hidden_dim=16, entities=ents, relations=rels, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=16, train_triples=train, entities=ents, relations=rels, seed=1)
        opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-2, lazy_rows=lazy)
        step = FusedTrainStep(m, 1.0)
        for it in range(3):
            for half in range(2):  # two backward passes per optimizer step
                s = t[(2 * it + half) * 32: (2 * it + half + 1) * 32].contiguous()
                step(s, w, ns.generate(s, "tail-batch"), "tail-batch")
            opt.step()
            opt.zero_grad()
            assert not m.entity_embedding.grad.any(), "stale gradient rows survived the step"
        if lazy:
            opt.flush()
        return m.entity_embedding.detach().clone()

    a, b = run(True), run(False)  # (fp32 atomics: the two runs agree to rounding, not bitwise)
    assert torch.allclose(a, b, rtol=0, atol=1e-6), float((a - b).abs().max())


def test_collapsed_or_diverged_model_does_not_rank_first():
    """evaluation.py:245-262 sorts the scores; a target tied with everything (collapsed model) or NaN (diverged model) sits
    somewhere in the pack there.  The device ranking must not report rank 1 for it (it used to: nothing compares greater);
    it reports the position in a stable descending sort: 1 + the unfiltered candidates with a smaller id."""
    from mkb_amd import datasets, evaluation, models

    ds = datasets.CountriesS1(batch_size=8, seed=42, num_workers=0)
    true = set(ds.true_triples)
    ranks = []
    for h, r, t in ds.test:  # the Evaluation walks the head-batch stream first, then the tail-batch stream
        ranks.append(1 + sum(1 for e in range(h) if (e, r, t) not in true))
    for h, r, t in ds.test:
        ranks.append(1 + sum(1 for e in range(t) if (h, r, e) not in true))
    ranks = np.asarray(ranks, dtype=np.float64)
    want = {"MRR": round(float((1 / ranks).mean()), 4), "MR": round(float(ranks.mean()), 4),
            "HITS@1": round(float((ranks <= 1).mean()), 4), "HITS@3": round(float((ranks <= 3).mean()), 4),
            "HITS@10": round(float((ranks <= 10).mean()), 4)}
    assert want["MR"] > 100 and want["HITS@1"] < 0.1
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=8,
                               device="cuda", num_workers=0)
    m = models.TransE(hidden_dim=8, entities=ds.entities, relations=ds.relations, gamma=3.0).cuda()
    for fill in (0.0, float("nan")):  # every candidate scores the same / every score is NaN
        with torch.no_grad():
            m.entity_embedding.fill_(fill)
            m.relation_embedding.fill_(fill)
        res = ev.eval(model=m, dataset=ds.test)
        assert res == pytest.approx(want, abs=1e-4), (fill, res, want)
        rel = ev.eval_relations(model=m, dataset=ds.test)
        assert rel["MR_relations"] >= 1.0 and rel["MRR_relations"] <= 1.0


def test_pipeline_on_device_batches_trains_and_matches_dataset_order_without_shuffle(capsys):
    """Pipeline.device_batches (opt-in): with shuffle=False the device producer yields the dataset's own
    batches, so the run must end with the very same tables as the default host producer."""
    from mkb_amd import compose, datasets, losses, models, optim, sampling

    def run(device_batches):
        ds = datasets.CountriesS1(batch_size=128, shuffle=False, seed=42, num_workers=0)
        torch.manual_seed(5)
        m = models.RotatE(hidden_dim=20, entities=ds.entities, relations=ds.relations, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=16, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
        opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-3)
        pipe = compose.Pipeline(epochs=2, eval_every=100, device="cuda")
        pipe.device_batches = device_batches
        pipe.learn(model=m, dataset=ds, sampling=ns, optimizer=opt, loss=losses.Adversarial(alpha=0.5),
                   evaluation=__import__("mkb_amd").evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities,
                                                                         relations=ds.relations, batch_size=64, device="cuda"))
        return m.entity_embedding.detach().clone(), pipe.test_scores

    a, sa = run(False)
    b, sb = run(True)
    assert torch.allclose(a, b, rtol=0, atol=1e-6), float((a - b).abs().max())  # (fp32 atomics: equal to rounding)
    assert sa == pytest.approx(sb, abs=1e-4)


def test_out_of_range_ids_raise_index_error_like_index_select():
    """models/base.py:166-207 gathers with index_select, which raises IndexError for an id outside the table.  The kernels
    index the tables directly: model(...) flags such ids on the device (mkb_check_ids) and check_ids() raises."""
    from mkb_amd import models

    N, R = 50, 3
    m = models.TransE(hidden_dim=8, entities={i: i for i in range(N)}, relations={i: i for i in range(R)}, gamma=3.0).cuda()
    good = torch.tensor([[1, 0, 2], [3, 2, 4]]).cuda()
    m(good)
    m(good, torch.tensor([[5, 6], [7, 49]]).cuda(), "tail-batch")
    m.check_ids()  # nothing flagged
    m(torch.tensor([[1, 3, 2]]).cuda()[:, [0, 1, 2]].clamp(max=N - 1))  # relation 3 does not exist
    with pytest.raises(IndexError, match="relation id"):
        m.check_ids()
    m.check_ids()  # the flag was cleared
    m(good[:1], torch.tensor([[0, N - 1]]).cuda(), "head-batch")
    m.check_ids()
    torch.cuda.synchronize()


@pytest.mark.parametrize("optimizer", ["torch", "row-lazy"])
def test_table_gradients_with_a_regulariser_on_both_tables(optimizer):
    """README-style step (positive and negative scores from two ``model(...)`` calls, one ``loss.backward()``) with a norm
    regulariser on BOTH tables in the same graph -- a third and fourth contributor to the tables' gradients besides the two score
    functions.  ``.grad`` must be scores' gradient + regulariser's gradient on every row (round 5's shared buffer lost the rows of
    whichever score function ran its backward after the engine had summed the first one's buffer with the regulariser's).  Both
    gradient routes: dense buffers handed to autograd (torch.optim) and rows added straight into ``.grad`` (row-lazy Adam); two
    passes without zero_grad in between, so ``.grad`` accumulates."""
    from mkb_amd import datasets, losses, models, optim, sampling

    ds = datasets.Umls(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    s, w = train[:64].contiguous(), (torch.rand(64) + 0.1).cuda()
    crit = losses.Adversarial(alpha=0.5)

    def grads(name, reg):
        torch.manual_seed(3)
        m = getattr(models, name)(hidden_dim=20, entities=ds.entities, relations=ds.relations, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=8, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=4)
        if optimizer == "row-lazy":
            opt = optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, lazy_rows=True)  # noqa: F841 (owns the tables)
        out = []
        for mode in ("head-batch", "tail-batch"):
            neg = ns.generate(s, mode)
            loss = crit(m(s), m(s, neg, mode), w)
            if reg:
                loss = loss + 1e-3 * (m.entity_embedding ** 2).sum() + 1e-3 * (m.relation_embedding.abs() ** 3).sum()
            loss.backward()
            out.append((m.entity_embedding.grad.clone(), m.relation_embedding.grad.clone(),
                        m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone()))
        return out

    for name in ("RotatE", "TransE", "pRotatE", "ComplEx"):
        a, b = grads(name, True), grads(name, False)
        for k, ((ea, ra, e, r), (eb, rb, _, _)) in enumerate(zip(a, b)):
            reg_e, reg_r = (k + 1) * 2e-3 * e, (k + 1) * 3e-3 * r * r.abs()  # (accumulated over k + 1 passes; tables do not move)
            np.testing.assert_allclose(ea.cpu().numpy(), (eb + reg_e).cpu().numpy(), rtol=0, atol=3e-7)
            np.testing.assert_allclose(ra.cpu().numpy(), (rb + reg_r).cpu().numpy(), rtol=0, atol=2e-6)


def test_multi_tensor_dense_adam_equals_one_launch_per_tensor_bit_for_bit():
    """``mkb_adam_step_multi`` (the dense tensors of a step in one launch, each with its own step count) against
    ``mkb_adam_step`` per tensor: the same scalars, the same element arithmetic, the same bits -- odd sizes included."""
    import ctypes

    from mkb_amd import _hip

    lib = _hip.lib()
    g = torch.Generator().manual_seed(7)
    sizes, steps = [135 * 64, 49 * 64 + 3, 1, 5000 * 36], [3, 1, 7, 2]

    def make():
        return [[torch.randn(n, generator=torch.Generator().manual_seed(100 + i + 10 * k)).cuda() for k in range(4)] for i, n in enumerate(sizes)]

    a, b = make(), make()
    for (p, gr, m, v), n, st in zip(a, sizes, steps):
        v.abs_()
        _hip.check(lib.mkb_adam_step(_hip.ptr(p), _hip.ptr(gr), _hip.ptr(m), _hip.ptr(v), n, st, 1e-2, 0.9, 0.999, 1e-8, 1, _hip.stream_ptr()), "single")
    for (p, gr, m, v) in b:
        v.abs_()
    arr = (_hip.AdamDense * len(b))(*[_hip.AdamDense(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, st)
                                       for (p, gr, m, v), n, st in zip(b, sizes, steps)])
    _hip.check(lib.mkb_adam_step_multi(arr, len(b), 1e-2, 0.9, 0.999, 1e-8, 1, None, _hip.stream_ptr()), "multi")
    for ta, tb in zip(a, b):
        for x, y in zip(ta, tb):
            assert torch.equal(x, y)
    assert not any(t[1].any().item() for t in b)  # zero_grad rode the launch


def test_out_of_range_entity_id_in_front_of_a_row_lazy_table_is_skipped_not_replayed():
    """model(...) in front of a row-lazily stepped table first makes the rows it is about to read current
    (optimizer.catch_up with the caller's ids).  An id >= n_entity used to reach the replay kernel, which exchanged last[row] and
    could rewrite p / m / v out of bounds before check_ids() had a chance to raise.  Now the replay skips such ids: the forward
    still flags them (IndexError from check_ids(), like the reference's index_select), and the run continues exactly as if the bad
    call had not happened."""
    from mkb_amd import losses, models, optim

    N, R, hid = 6000, 5, 16
    g = torch.Generator().manual_seed(1)
    samples = [torch.stack([torch.randint(N, (64,), generator=g), torch.randint(R, (64,), generator=g),
                            torch.randint(N, (64,), generator=g)], 1).cuda() for _ in range(3)]
    negs = [torch.randint(N, (64, 8), generator=g).cuda() for _ in range(3)]
    w = torch.ones(64).cuda()
    crit = losses.Adversarial(alpha=0.5)

    def run(with_bad_call):
        torch.manual_seed(2)
        m = models.RotatE(hidden_dim=hid, entities={i: i for i in range(N)}, relations={i: i for i in range(R)}, gamma=6.0).cuda()
        opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-2, lazy_rows=True)
        for i in range(3):
            if with_bad_call and i == 2:
                bad = negs[i].clone()
                bad[0, 0], bad[5, 3] = N + 7, N  # past the table
                with torch.no_grad():
                    m(samples[i], bad, "tail-batch")
                with pytest.raises(IndexError):
                    m.check_ids()
            crit(m(samples[i]), m(samples[i], negs[i], "tail-batch"), w).backward()
            opt.step()
            opt.zero_grad()
        opt.flush()
        m.check_ids()
        return m.entity_embedding.detach().clone(), {k: v.clone() for k, v in opt.state[m.entity_embedding].items() if torch.is_tensor(v)}

    a, sa = run(False)
    b, sb = run(True)
    # (the gradients of model(...) are summed with fp32 atomics, so two runs agree to rounding and Adam's scale-free update turns a
    # last-bit difference of a near-zero gradient element into up to one step of lr: the tables are compared within two steps, the
    # integer bookkeeping -- which step every row is current through -- exactly)
    assert torch.equal(sa["last"], sb["last"])
    assert float((a - b).abs().max()) <= 2.5e-2 and float(((a - b).abs() > 1e-5).float().mean()) < 0.02
    for k in ("m", "v"):
        assert sa[k].shape == sb[k].shape and torch.isfinite(sb[k]).all() and torch.allclose(sa[k], sb[k], rtol=0, atol=1e-3), k

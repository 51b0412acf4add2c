"""``mkb_amd.optim.Adam(lazy_rows=True, defer_step=True)``: the real step of the touched rows waits in their gradient rows
until the next catch-up / flush visits them (mkb_adam_rows_advance*, mkb_amd/csrc/adam.hip).  Same arithmetic in the
same order as the separate step launch, so everything here is compared BIT FOR BIT against ``defer_step=False`` wherever
the gradients themselves are deterministic, and against dense Adam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tables(seed=5, n=5000, d=64, r=37):
    g = torch.Generator(device="cpu").manual_seed(seed)
    ent = torch.nn.Parameter(torch.randn(n, d, generator=g).cuda())  # >= 4096 rows: steps row-lazily
    rel = torch.nn.Parameter(torch.randn(r, d, generator=g).cuda())  # small: dense, rides the row launch
    ent.grad, rel.grad = torch.zeros_like(ent), torch.zeros_like(rel)
    return ent, rel


def _synthetic_steps(opt, ent, rel, steps, seed=11, on_step=None):
    """Hand-made gradients through the optimizer's own protocol: catch-up of the rows, THEN their gradient, then step."""
    from mkb_amd import _links

    g = torch.Generator(device="cpu").manual_seed(seed)
    for it in steps:
        ids = torch.randint(ent.shape[0], (300,), generator=g).cuda()  # with duplicates, like a batch's pool | heads | tails
        vals = torch.randn(300, ent.shape[1], generator=g).cuda()
        relg = torch.randn(rel.shape, generator=g).cuda()
        if it == 5:
            opt.lr = 3e-3  # a scheduler changing the rate: the deferred step must use the rate of ITS step
        opt.catch_up(ent, ids)
        uniq = torch.unique(ids)
        ent.grad[uniq] += vals[: uniq.numel()]  # one deterministic value per distinct row
        rel.grad += relg
        _links.mark_touched(ent, ids)
        opt.step()
        opt.zero_grad()
        if on_step is not None:
            on_step(it)


@pytest.mark.parametrize("d", [64, 33])
def test_deferred_step_equals_separate_step_launch_bit_for_bit(d):
    from mkb_amd import optim

    out = []
    for defer in (True, False):
        ent, rel = _tables(d=d)
        opt = optim.Adam([ent, rel], lr=1e-2, lazy_rows=True, defer_step=defer)
        mids = []

        def peek(it):
            if it == 7:  # an evaluation in the middle of training: flush, read, carry on
                opt.flush()
                mids.append((ent.detach().clone(), rel.detach().clone()))

        _synthetic_steps(opt, ent, rel, range(12), on_step=peek)
        if defer:
            assert opt.state[ent].get("defer") and ent.grad.any(), "the last step should still be waiting in the gradient rows"
        opt.flush()
        assert not ent.grad.any() and not rel.grad.any()
        st = opt.state[ent]
        out.append((mids[0][0], mids[0][1], ent.detach().clone(), rel.detach().clone(), st["m"].clone(), st["v"].clone(),
                    opt.state[rel]["m"].clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    # ... and dense Adam on the same gradients (torch.optim.Adam, a few ulp of the update)
    ent, rel = _tables(d=d)
    ref = torch.optim.Adam([ent, rel], lr=1e-2)

    class Dense:  # the protocol of _synthetic_steps on top of torch.optim
        lr = 1e-2

        def catch_up(self, *a): pass

        def step(self):
            for grp in ref.param_groups:
                grp["lr"] = self.lr
            ref.step()

        def zero_grad(self):
            ent.grad.zero_(); rel.grad.zero_()

    _synthetic_steps(Dense(), ent, rel, range(12))
    np.testing.assert_allclose(out[0][2].cpu().numpy(), ent.detach().cpu().numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(out[0][3].cpu().numpy(), rel.detach().cpu().numpy(), rtol=0, atol=3e-6)


@pytest.mark.parametrize("defer", [True, False])
def test_sweep_window_of_the_per_step_launch_changes_nothing(defer, monkeypatch):
    """The per-step optimizer launch may also visit a moving window of table rows (adam.hip, "Sweep": keeps the pending
    lists short on tables whose entities are rarely touched).  When a row is replayed does not change what is replayed:
    tables, moments and a mid-run flush are bit-identical for every window period, with and without the deferred step."""
    from mkb_amd import optim

    out = []
    for period in ("0", "1", "3", "64", None):  # off, the whole table every step, 1/3 of it, 1/64, the built-in rule
        if period is None:
            monkeypatch.delenv("MKB_ADAM_SWEEP", raising=False)
        else:
            monkeypatch.setenv("MKB_ADAM_SWEEP", period)
        ent, rel = _tables(n=20000, d=36)  # 20000 >= 32 x 300 listed rows: the built-in rule sweeps here
        opt = optim.Adam([ent, rel], lr=1e-2, lazy_rows=True, defer_step=defer)
        mids = []

        def peek(it):
            if it == 9:
                opt.flush()
                mids.append(ent.detach().clone())

        _synthetic_steps(opt, ent, rel, range(15), on_step=peek)
        last = opt.state[ent]["last"].clone()
        opt.flush()
        st = opt.state[ent]
        out.append((mids[0], ent.detach().clone(), rel.detach().clone(), st["m"].clone(), st["v"].clone()))
        if period == "1":  # every row was visited by the last launch: nothing older than the step before the last
            assert int(last.min()) >= 13
        if period == "0":
            assert int(last.min()) <= 10  # (without the window most rows were last visited by the flush after step 9)
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert torch.equal(a, b)


def test_deferred_step_survives_set_to_none_and_checkpoint_resume():
    from mkb_amd import optim

    def run(defer, resume_at=None):
        ent, rel = _tables()
        opt = optim.Adam([ent, rel], lr=1e-2, lazy_rows=True, defer_step=defer)
        _synthetic_steps(opt, ent, rel, range(4))
        opt.zero_grad(set_to_none=True)  # really clears: whatever is waiting must be applied first
        assert ent.grad is None
        ent.grad, rel.grad = torch.zeros_like(ent), torch.zeros_like(rel)
        _synthetic_steps(opt, ent, rel, range(4, 8), seed=12)
        if resume_at is not None:
            sd = opt.state_dict()  # flushes
            e2, r2 = _tables()
            with torch.no_grad():
                e2.copy_(ent); r2.copy_(rel)
            opt2 = optim.Adam([e2, r2], lr=1e-2, lazy_rows=True, defer_step=defer)
            opt2.load_state_dict(sd)
            ent, rel, opt = e2, r2, opt2
        _synthetic_steps(opt, ent, rel, range(8, 12), seed=13)
        opt.flush()
        return ent.detach().clone(), rel.detach().clone()

    want = run(False)
    for got in (run(True), run(True, resume_at=8)):
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_gradient_rows_without_catch_up_are_refused_when_deferring():
    from mkb_amd import _links, optim

    ent, rel = _tables()
    opt = optim.Adam([ent, rel], lr=1e-2, lazy_rows=True, defer_step=True)
    _synthetic_steps(opt, ent, rel, range(3))
    ids = torch.arange(10, device="cuda")
    ent.grad[ids] += 1.0          # no catch-up of these rows in this step
    _links.mark_touched(ent, ids)
    with pytest.raises(RuntimeError, match="stop_deferring"):
        opt.step()


def test_fused_training_with_deferred_step_matches_the_separate_launch():
    """The whole fused loop (sampler riding the catch-up, gradient accumulation, a plain ``model(...)`` call and an
    evaluation in between, which flush) with and without ``defer_step``; the gradients come from fp32 atomics, so the
    two runs agree to rounding."""
    from mkb_amd import models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    N, R = 5000, 4
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    rs = np.random.RandomState(1)
    train = np.stack([rs.randint(N, size=3000), rs.randint(R, size=3000), rs.randint(N, size=3000)], 1)
    t = torch.as_tensor(train).cuda()
    w = torch.ones(32, device="cuda")

    def run(defer, name):
        torch.manual_seed(3)
        m = getattr(models, name)(hidden_dim=16, entities=ents, relations=rels, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=16, train_triples=train, entities=ents, relations=rels, seed=1)
        params = [m.entity_embedding, m.relation_embedding] + ([m.modulus] if name == "pRotatE" else [])
        opt = optim.Adam(params, lr=1e-2, lazy_rows=True, draw_ahead=ns, defer_step=defer)
        step = FusedTrainStep(m, 1.0)
        probes = []
        for it in range(10):
            mode = "tail-batch" if it % 2 else "head-batch"
            for half in range(2 if it in (3, 4) else 1):  # steps 3 and 4 accumulate two backward passes
                s = t[(2 * it + half) * 32: (2 * it + half + 1) * 32].contiguous()
                step.sampled(s, w, ns, mode)
            opt.step()
            opt.zero_grad()
            if it == 6:
                with torch.no_grad():
                    probes.append(m(t[:8].contiguous()).clone())  # general path: flushes what is pending first
        opt.flush()
        return [m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone()] + probes + \
            ([m.modulus.detach().clone()] if name == "pRotatE" else []), opt

    for name in ("TransE", "RotatE", "pRotatE"):
        (a, oa), (b, ob) = run(True, name), run(False, name)
        assert oa.state[oa.params[0]].get("defer") and not ob.state[ob.params[0]].get("defer")
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=0, atol=2e-6), (name, float((x - y).abs().max()))


def test_pipeline_turns_the_deferred_step_on_and_trains_the_same_model():
    from mkb_amd import compose, datasets, evaluation, losses, models, optim, sampling

    def run(defer):
        ds = datasets.Fb15k237(batch_size=2048, shuffle=True, seed=42, num_workers=0)
        db = datasets.DeviceBatches(ds, "cuda", seed=42)
        torch.manual_seed(42)
        m = models.TransE(hidden_dim=16, entities=ds.entities, relations=ds.relations, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=8, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
        opt = optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, lazy_rows=True, defer_step=defer)
        ds.valid, ds.test = [], []
        pipe = compose.Pipeline(epochs=1, eval_every=100, device="cuda")
        pipe.learn(model=m, dataset=db, sampling=ns, optimizer=opt, loss=losses.Adversarial(alpha=1.0))
        assert not m.entity_embedding.grad.any(), "learn() must leave no deferred step behind"
        return m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(), opt, pipe.metric_loss.get()

    e1, r1, o1, l1 = run(None)
    e0, r0, o0, l0 = run(False)
    # the deferral is scoped to learn(): given back on return, nothing pending, step() applies steps again (torch.optim
    # semantics for whatever the user's own code does next, e.g. model.zero_grad())
    assert o1.defer_step is None and o0.defer_step is False and o1.draw_ahead is None
    assert not any(st.get("defer") for st in o1.state.values())
    assert abs(l1 - l0) < 1e-4
    # The gradients come from fp32 atomics: two runs of the SAME configuration either agree to ~5e-7 or -- when one
    # L1 sign sum cancels to an exact 0 in one order and to a rounding residue in the other -- differ by an lr-sized step in
    # a handful of elements (Adam normalises the residue to +-1; seen in 2 of 5 identical runs).  So: nearly all elements
    # agree tightly, none by more than a few steps' worth.
    for a, b in ((e1, e0), (r1, r0)):
        diff = (a - b).abs()
        assert int((diff > 2e-5).sum()) <= 32 and float(diff.max()) < 1e-2, (int((diff > 2e-5).sum()), float(diff.max()))


def test_user_code_after_learn_may_clear_gradients_any_way_it_likes():
    """After learn() the optimizer is back to plain semantics: a hand-written step that clears gradients with
    model.zero_grad(set_to_none=True) must lose nothing (with a pending deferred step it would silently drop one)."""
    from mkb_amd import compose, datasets, losses, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    def run(borrow):
        ds = datasets.Fb15k237(batch_size=2048, shuffle=True, seed=42, num_workers=0)
        db = datasets.DeviceBatches(ds, "cuda", seed=42)
        torch.manual_seed(42)
        m = models.TransE(hidden_dim=16, entities=ds.entities, relations=ds.relations, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=8, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
        opt = optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, lazy_rows=True, defer_step=None if borrow else False)
        ds.valid, ds.test = [], []
        compose.Pipeline(epochs=1, eval_every=100, device="cuda").learn(model=m, dataset=db, sampling=ns, optimizer=opt,
                                                                        loss=losses.Adversarial(alpha=1.0))
        before = m.entity_embedding.detach().clone()
        step = FusedTrainStep(m, 1.0)
        s = torch.as_tensor(np.asarray(ds.train[:512], dtype=np.int64)).cuda()
        w = torch.ones(512, device="cuda")
        for _ in range(2):  # the user's own loop
            step(s, w, ns.generate(s, "tail-batch"), "tail-batch")
            opt.step()
            m.zero_grad(set_to_none=True)
        opt.flush()
        moved = (m.entity_embedding.detach() - before).abs().max().item()
        return moved

    a, b = run(True), run(False)
    assert a > 0 and abs(a - b) < 1e-6, (a, b)


def test_autograd_backward_before_a_fused_step_in_one_optimizer_step_is_not_overwritten():
    """with ``defer_step`` on, an autograd backward (``model(...)`` + ``loss.backward()``) earlier in the SAME
    optimizer step accumulates into ``.grad`` without marking rows; the fused step behind it must then accumulate
    (``rows_clear`` off) instead of storing over those rows, and the optimizer must step every written row.  Compared with
    ``torch.optim.Adam`` on a twin model that takes both halves through autograd and the general kernels."""
    from mkb_amd import losses, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep
    import mkb_amd.models.base as model_base

    N, R = 5000, 4
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    rs = np.random.RandomState(1)
    train = np.stack([rs.randint(N, size=3000), rs.randint(R, size=3000), rs.randint(N, size=3000)], 1)
    t = torch.as_tensor(train).cuda()
    w = torch.ones(32, device="cuda")
    crit = losses.Adversarial(alpha=1.0)

    def run(fast):
        torch.manual_seed(3)
        m = models.RotatE(hidden_dim=16, entities=ents, relations=rels, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=16, train_triples=train, entities=ents, relations=rels, seed=1)
        if fast:
            opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-2, lazy_rows=True, defer_step=True)
        else:
            opt = torch.optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-2)
        step = FusedTrainStep(m, 1.0)
        for it in range(6):
            a = t[(2 * it) * 32: (2 * it + 1) * 32].contiguous()
            b = t[(2 * it + 1) * 32: (2 * it + 2) * 32].contiguous()
            mixed = it in (2, 3, 5)
            if mixed or not fast:  # first half through autograd (for the twin: always)
                neg = ns.generate(a, "head-batch")
                crit(m(a), m(a, neg.clone() if not fast else neg, "head-batch"), w).backward()
            else:
                step(a, w, ns.generate(a, "head-batch"), "head-batch")
            neg = ns.generate(b, "tail-batch")
            if fast:
                step(b, w, neg, "tail-batch")
            else:
                crit(m(b), m(b, neg.clone(), "tail-batch"), w).backward()
            opt.step()
            opt.zero_grad()
        if fast:
            opt.flush()
        return m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone()

    old = model_base.AUTO_POOL
    try:
        model_base.AUTO_POOL = False  # the twin really takes the general kernels
        ref = run(False)
    finally:
        model_base.AUTO_POOL = old
    got = run(True)
    for x, y in zip(got, ref):
        assert torch.allclose(x, y, rtol=0, atol=5e-6), float((x - y).abs().max())


@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("name", ["RotatE", "TransE", "RotatE-plain"])
def test_readme_loop_keeps_the_row_lazy_route(name, defer, monkeypatch):
    """The reference's own loop (README.md:448-474: ``model(sample)``, ``model(sample, negatives, mode)``, ``loss.backward()``,
    ``optimizer.step()``) with ``mkb_amd.optim.Adam(lazy_rows=True)``: the backward functions add their rows straight into
    ``.grad`` of the row-lazily stepped table and record them (``_gradshare.direct``), the forward passes bring the rows they
    read up to date -- no dense gradient goes through autograd, and the optimizer must STILL be on its row-lazy route at the
    end (it used to fall back to the dense kernel after the first step).  Same tables as ``torch.optim.Adam`` on a twin."""
    from mkb_amd import _links, losses, models, optim, sampling
    import mkb_amd.models.base as model_base

    plain = name.endswith("-plain")  # negatives without the sampler's pool description: the general kernels, candidate by candidate
    name = name.split("-")[0]
    if plain:
        monkeypatch.setattr(model_base, "AUTO_POOL", False)
    N, R = 6000, 5
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    rs = np.random.RandomState(2)
    train = np.stack([rs.randint(N, size=4000), rs.randint(R, size=4000), rs.randint(N, size=4000)], 1)
    t = torch.as_tensor(train).cuda()
    w = torch.rand(64, device="cuda") + 0.5
    crit = losses.Adversarial(alpha=0.5)

    def run(fast):
        torch.manual_seed(4)
        m = getattr(models, name)(hidden_dim=24, entities=ents, relations=rels, gamma=6.0).cuda()
        ns = sampling.NegativeSampling(size=32, train_triples=train, entities=ents, relations=rels, seed=3)
        ps = [m.entity_embedding, m.relation_embedding]
        opt = optim.Adam(ps, lr=1e-2, lazy_rows=True, defer_step=defer) if fast else torch.optim.Adam(ps, lr=1e-2)
        seen = []
        for it in range(9):
            s = t[it * 64: (it + 1) * 64].contiguous()
            mode = "head-batch" if it % 2 == 0 else "tail-batch"
            opt.zero_grad()
            pos = m(s)
            neg = ns.generate(s, mode)
            if plain:
                neg = neg.clone()
            err = crit(pos, m(s, neg, mode), w)
            err.backward()
            if fast and it == 4:  # a second backward pass before one step (gradient accumulation)
                crit(m(s), m(s, neg, mode), w).backward()
            elif it == 4:
                crit(m(s), m(s, neg, mode), w).backward()
            opt.step()
            seen.append(float(err.detach()))
            if it == 6:  # evaluation in the middle: scores of triples the loop has not touched must see current rows
                with torch.no_grad():
                    seen.append(float(m(t[2000:2064].contiguous()).sum()))
        if fast:
            st = opt.state[m.entity_embedding]
            assert "last" in st and _links.owner(m.entity_embedding) is opt, "the optimizer fell back to the dense kernel"
            assert not _links.autograd_wrote(m.entity_embedding)
            opt.flush()
        return m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(), seen

    ref = run(False)
    got = run(True)
    # (both runs add gradient rows with fp32 atomics: where the contributions to an element nearly cancel, their order moves the
    # Adam update of that element by a few per cent of lr -- seen: 5.1e-5 at lr 1e-2, in one process out of two.  A row that
    # missed a step or was read stale is off by ~lr per step: two orders of magnitude above the bound.)
    assert np.allclose(got[2], ref[2], rtol=1e-4, atol=1e-4), (got[2], ref[2])
    for x, y in zip(got[:2], ref[:2]):
        assert torch.allclose(x, y, rtol=0, atol=3e-4), float((x - y).abs().max())
        assert float((x - y).abs().mean()) < 2e-6, float((x - y).abs().mean())

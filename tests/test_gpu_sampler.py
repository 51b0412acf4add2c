"""-m gpu: the on-device sampler must be BIT-EXACT with the reference (golden negatives from the live
reference + the numpy/C oracle on seeded batches)."""
import numpy as np
import pytest
import torch

from mkb_amd import _links

pytestmark = pytest.mark.gpu


def test_reference_doctest_known_answers(golden):
    """sampling/negative_sampling.py:101-103, 120-122 (size=5, seed 42, 4-entity toy graph)."""
    from mkb_amd import sampling

    ents, rels = {i: i for i in range(4)}, {i: i for i in range(4)}
    train = [(0, 0, 1), (1, 0, 2), (2, 0, 3), (3, 0, 1)]
    ns = sampling.NegativeSampling(size=5, train_triples=train, entities=ents, relations=rels, seed=42)
    smp = torch.tensor([[0, 0, 1], [1, 0, 2]]).cuda()
    tail = ns.generate(smp, mode="tail-batch")
    head = ns.generate(smp, mode="head-batch")
    assert tail.dtype == torch.int64 and tail.shape == (2, 5)
    np.testing.assert_array_equal(tail.cpu().numpy(), [[2, 3, 0, 2, 2], [3, 0, 3, 0, 0]])
    np.testing.assert_array_equal(head.cpu().numpy(), [[2, 2, 2, 2, 2], [2, 2, 2, 2, 3]])


@pytest.mark.parametrize("cls", ["Umls", "Wn18rr", "Fb15k237"])
def test_real_graph_negatives_bit_exact(golden, cls):
    from mkb_amd import datasets, sampling

    g = golden("sampler.npz")
    K = int(g[f"{cls}/K"])
    ds = getattr(datasets, cls)(batch_size=8, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    c = 0
    while f"{cls}/{c}/idx" in g.files:
        smp = train[torch.as_tensor(g[f"{cls}/{c}/idx"].astype(np.int64)).cuda()]
        mode = "head-batch" if c % 2 == 0 else "tail-batch"
        neg = ns.generate(smp, mode)
        np.testing.assert_array_equal(neg.cpu().numpy(), g[f"{cls}/{c}/neg"].astype(np.int64))
        info = neg._mkb_pool
        # side outputs are consistent: neg == pool[pos]; cnt == histogram of pos
        np.testing.assert_array_equal(info.pool[info.pos.long()].cpu().numpy(), neg.cpu().numpy())
        hist = torch.zeros(neg.shape[0], 2 * K, dtype=torch.int64, device="cuda")
        hist.scatter_add_(1, info.pos.long(), torch.ones_like(info.pos, dtype=torch.int64))
        np.testing.assert_array_equal(info.cnt.cpu().numpy().astype(np.int64), hist.cpu().numpy())
        c += 1
    ns.check()
    assert c >= 4


def test_long_stream_vs_c_oracle(liboracle):
    """Headline shape: FB15k-237, K=256, B=1024, 40 consecutive batches alternating modes (the MT19937 block
    boundary is crossed many times) vs the plain-C restatement."""
    import ctypes

    from mkb_amd import datasets, sampling
    from mkb_amd.sampling.negative_sampling import _filter_csr

    ds = datasets.Fb15k237(batch_size=8, shuffle=False, seed=42, num_workers=0)
    train_np = np.asarray(ds.train, dtype=np.int64)
    train = torch.as_tensor(train_np).cuda()
    K, B, N, R = 256, 1024, ds.n_entity, ds.n_relation
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=7)
    (hk, ho, hv, _), (tk, to, tv, _) = _filter_csr(ds.train, N, R)
    st = ctypes.create_string_buffer(4 * 624 + 4)
    liboracle.orc_mt_seed(st, ctypes.c_uint32(7))
    liboracle.orc_generate.restype = ctypes.c_int
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    pick = np.random.RandomState(3)
    for c in range(40):
        idx = pick.randint(len(train_np), size=B)
        head = c % 2 == 0
        got = ns.generate(train[torch.as_tensor(idx).cuda()], "head-batch" if head else "tail-batch")
        smp = np.ascontiguousarray(train_np[idx])
        want = np.zeros((B, K), dtype=np.int64)
        pool = np.zeros(2 * K, dtype=np.int64)
        k, o, v, stride = (hk, ho, hv, N) if head else (tk, to, tv, R)
        rc = liboracle.orc_generate(st, ctypes.c_int64(N), ctypes.c_int64(K), p(smp), ctypes.c_int64(B), ctypes.c_int(head),
                                    p(k), ctypes.c_int64(len(k)), p(o), p(v), ctypes.c_int64(stride), p(want), p(pool))
        assert rc == 0
        np.testing.assert_array_equal(got._mkb_pool.pool.cpu().numpy(), pool)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    key, pos = ns.get_state()
    ref = np.frombuffer(st.raw[: 4 * 624], dtype=np.uint32)
    np.testing.assert_array_equal(key, ref)
    assert pos == int(np.frombuffer(st.raw[4 * 624:], dtype=np.int32)[0])


def test_unseen_key_raises_keyerror():
    from mkb_amd import sampling

    ents, rels = {i: i for i in range(4)}, {i: i for i in range(2)}
    ns = sampling.NegativeSampling(size=3, train_triples=[(0, 0, 1), (1, 0, 2)], entities=ents, relations=rels, seed=1)
    ns.generate(torch.tensor([[0, 0, 1], [3, 1, 0]]).cuda(), "tail-batch")
    with pytest.raises(KeyError):
        ns.check()


def test_empty_filter_raises_instead_of_hanging():
    from mkb_amd import sampling

    ents, rels = {i: i for i in range(2)}, {0: 0}
    ns = sampling.NegativeSampling(size=3, train_triples=[(0, 0, 0), (0, 0, 1)], entities=ents, relations=rels, seed=1)
    ns.generate(torch.tensor([[0, 0, 1]]).cuda(), "tail-batch")  # true tails of (0,0) = {0,1} = every entity
    with pytest.raises(RuntimeError, match="whole candidate pool"):
        ns.check()


def test_empty_batch_behaves_like_the_reference():
    """What the reference answers an empty batch with (probed on the live import, CountriesS1 / RotatE hidden 4): a positive
    forward -> an empty [0, 1] score; a negative forward -> RuntimeError (view of zero elements); ``generate`` -> RuntimeError
    (``torch.stack`` of no rows) AFTER the batch's pool was drawn: the next call sees the stream's second pool; the loss of no
    rows -> nan (0 / 0).  None of it launches a kernel."""
    from mkb_amd import datasets, losses, models, sampling

    ds = datasets.CountriesS1(batch_size=4, seed=42)
    m = models.RotatE(hidden_dim=4, entities=ds.entities, relations=ds.relations, gamma=3).cuda()
    s0 = torch.zeros((0, 3), dtype=torch.long).cuda()
    out = m(s0)
    assert tuple(out.shape) == (0, 1)
    out.sum().backward()  # differentiable, adds nothing
    assert float(m.entity_embedding.grad.abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        m(s0, torch.zeros((0, 5), dtype=torch.long).cuda(), mode="head-batch")
    assert torch.isnan(losses.Adversarial()(torch.zeros(0, 1).cuda(), torch.zeros(0, 5).cuda(), torch.zeros(0).cuda())).item()

    mk = lambda: sampling.NegativeSampling(size=5, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    batch = torch.as_tensor(np.asarray(ds.train[:6], dtype=np.int64)).cuda()
    a, b = mk(), mk()
    with pytest.raises(RuntimeError, match="non-empty"):
        a.generate(s0, "head-batch")
    b.generate(batch, "head-batch")               # b's first pool = the one a's empty call consumed
    na, nb = a.generate(batch, "tail-batch"), b.generate(batch, "tail-batch")
    np.testing.assert_array_equal(na.cpu().numpy(), nb.cpu().numpy())
    rs = np.random.RandomState(42)
    rs.randint(len(ds.entities), size=10)
    np.testing.assert_array_equal(na._mkb_pool.pool.cpu().numpy(), rs.randint(len(ds.entities), size=10))


def test_pool_drawn_ahead_inside_the_optimizer_launch_is_bit_identical():
    """mkb_adam_rows_catchup(draw_ahead=sampler): the next pool is drawn by one more workgroup of the optimizer's
    catch-up launch instead of the stand-alone kernel.  Negatives, pools and the reported generator state must
    equal those of a sampler that draws at generate() time, step after step; set_state discards a pool drawn ahead."""
    from mkb_amd import datasets, optim, sampling

    ds = datasets.Umls(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    mk = lambda: sampling.NegativeSampling(size=24, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=9)
    plain, ahead = mk(), mk()
    g = torch.Generator(device="cpu").manual_seed(1)
    ent = torch.nn.Parameter(torch.randn(5000, 64, generator=g).cuda())  # >= 4096 rows: steps row-lazily
    opt = optim.Adam([ent], lr=1e-3, lazy_rows=True, draw_ahead=ahead)
    for it in range(7):
        s = train[it * 64: (it + 1) * 64].contiguous()
        mode = "head-batch" if it % 2 else "tail-batch"
        a, b = plain.generate(s, mode), ahead.generate(s, mode)
        assert torch.equal(a, b), it
        assert torch.equal(a._mkb_pool.pool, b._mkb_pool.pool) and torch.equal(a._mkb_pool.cnt, b._mkb_pool.cnt)
        assert torch.equal(a._mkb_pool.touched, b._mkb_pool.touched)
        ids = torch.randperm(5000, generator=g)[:200].cuda()
        ent.grad = torch.zeros_like(ent)
        ent.grad[ids] = 1.0
        opt.catch_up(ent, ids)
        _links.mark_touched(ent, ids)
        opt.step()  # (the catch_up above carried the next pool's draw once a step had been taken)
        ka, pa = plain.get_state()
        kb, pb = ahead.get_state()   # the state BEFORE the pool drawn ahead
        assert pa == pb and np.array_equal(ka, kb), it
    ahead.set_state(*plain.get_state())  # discards the pool drawn ahead
    s = train[:64].contiguous()
    assert torch.equal(plain.generate(s, "tail-batch"), ahead.generate(s, "tail-batch"))
    plain.check(), ahead.check()


@pytest.mark.parametrize("size", [24, 512])
def test_sampler_riding_the_catch_up_launch_is_bit_identical(size):
    """mkb_adam_rows_catchup_generate: draw-ahead + this batch's filter + the row catch-up in one launch.  Every output
    of generate (negatives, pool, position map, multiplicities, touched rows) and the optimizer's tables must equal the
    separate calls', step after step, including the first step (nothing pending, pool not drawn ahead yet)."""
    from mkb_amd import datasets, optim, sampling

    ds = datasets.Umls(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    if size > 100:  # a sparse graph so that a 1024-candidate pool is never filtered empty
        ds = datasets.Wn18rr(batch_size=64, shuffle=False, seed=42, num_workers=0)
        train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    mk = lambda: sampling.NegativeSampling(size=size, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=9)
    tables = []
    for ride in (False, True):
        sam = mk()
        g = torch.Generator(device="cpu").manual_seed(1)
        ent = torch.nn.Parameter(torch.randn(max(5000, len(ds.entities)), 64, generator=g).cuda())  # >= 4096 rows: row-lazy
        opt = optim.Adam([ent], lr=1e-3, lazy_rows=True)
        outs = []
        for it in range(7):
            s = train[it * 61: it * 61 + 61].contiguous()  # 61 rows: the last filter workgroup is ragged
            mode = "head-batch" if it % 2 else "tail-batch"
            if ride:
                neg = sam.generate_with_catch_up(s, mode, opt, ent)
            else:
                neg = sam.generate(s, mode)
                opt.catch_up(ent, neg._mkb_pool.touched)
            info = neg._mkb_pool
            outs.append((neg.clone(), info.pool.clone(), info.pos.clone(), info.cnt.clone(), info.touched.clone()))
            ids = info.touched
            ent.grad = torch.zeros_like(ent)
            ent.grad[torch.unique(ids)] = 0.5
            _links.mark_touched(ent, ids)
            opt.step()
        opt.flush()
        sam.check()
        tables.append((outs, ent.detach().clone(), sam.get_state()))
    for a, b in zip(tables[0][0], tables[1][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(tables[0][1], tables[1][1])
    (ka, pa), (kb, pb) = tables[0][2], tables[1][2]
    assert pa == pb and np.array_equal(ka, kb)  # the riding sampler holds a pool drawn ahead: same logical state


def test_rocrand_pool_draw_is_uniform_filtered_and_resumable():
    """rng="rocrand" (opt-in, NOT the reference's numbers): the pool comes from rocRAND's Philox4x32-10 inside the draw
    kernel; everything else is the reference's algorithm.  Checked: ids in range, one shared pool per batch (<= 2K distinct
    negatives), every row's negatives avoid its true set and follow the first-K-survivors-cyclically rule against THAT
    pool, the draw is roughly uniform, differs from the numpy stream, is reproducible at a seed, and resumes exactly."""
    from mkb_amd import datasets, sampling

    ds = datasets.Fb15k237(batch_size=256, shuffle=False, seed=42, num_workers=0)
    train = np.asarray(ds.train, dtype=np.int64)
    K, B = 64, 256
    s = torch.as_tensor(train[np.random.RandomState(1).randint(len(train), size=B)]).cuda()

    def make(rng, seed=7):
        return sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=seed, rng=rng)

    a, b, ref = make("rocrand"), make("rocrand"), make("numpy")
    outs = []
    for mode in ("head-batch", "tail-batch", "head-batch"):
        na, nb, nr = a.generate(s, mode), b.generate(s, mode), ref.generate(s, mode)
        a.check()
        assert torch.equal(na, nb) and not torch.equal(na, nr)
        info = na._mkb_pool
        pool = info.pool.cpu().numpy()
        assert pool.min() >= 0 and pool.max() < len(ds.entities) and len(np.unique(na.cpu().numpy())) <= 2 * K
        true = a.true_head if mode == "head-batch" else a.true_tail
        neg = na.cpu().numpy()
        for i in range(0, B, 17):  # the reference's rule against this pool: survivors of the filter, cyclically, first K
            h, r, t = (int(v) for v in s[i].cpu())
            rec = true[(r, t)] if mode == "head-batch" else true[(h, r)]
            keep = pool[np.isin(pool, rec, assume_unique=True, invert=True)]
            want = np.concatenate([keep] * (K // max(1, len(keep)) + 1))[:K]
            np.testing.assert_array_equal(neg[i], want)
        outs.append(pool)
    allp = np.concatenate(outs)
    assert abs(allp.mean() / (len(ds.entities) - 1) - 0.5) < 0.08 and len(np.unique(allp)) > 0.9 * len(allp) * (1 - len(allp) / len(ds.entities))
    # resume: a fresh sampler set to a's state continues with a's next pool
    c = make("rocrand", seed=1)
    c.set_state(*a.get_state())
    assert torch.equal(a.generate(s, "tail-batch"), c.generate(s, "tail-batch"))


def test_rocrand_sampler_rides_the_fused_training_loop():
    """rng="rocrand" through the production loop (sampler filter + next-pool draw riding the row-lazy optimizer's launch,
    fused step, deferred Adam): runs, is reproducible at a seed, and a different seed trains on different negatives."""
    from mkb_amd import datasets, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    ds = datasets.Fb15k237(batch_size=128, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()

    def run(seed):
        torch.manual_seed(3)
        m = models.RotatE(hidden_dim=16, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
        ns = sampling.NegativeSampling(size=32, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=seed, rng="rocrand")
        opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-3, lazy_rows=True, draw_ahead=ns, defer_step=True)
        step = FusedTrainStep(m, alpha=1.0)
        w = torch.ones(128, device="cuda")
        losses = []
        for i in range(6):
            s = train[i * 128: (i + 1) * 128]
            losses.append(step.sampled(s, w, ns, "head-batch" if i % 2 == 0 else "tail-batch").item())
            opt.step()
            opt.zero_grad()
        opt.flush()
        ns.check()
        return losses, m.entity_embedding.detach().clone()

    (l1, e1), (l2, e2), (l3, e3) = run(5), run(5), run(6)
    # (same negatives, same losses; the tables agree to the order of the fp32 atomics of shared gradient rows)
    np.testing.assert_allclose(l1, l2, rtol=0, atol=1e-6)
    assert torch.allclose(e1, e2, rtol=0, atol=1e-5)
    assert l1 != l3 and all(np.isfinite(l1))


def test_rocrand_set_state_discards_a_pool_drawn_ahead():
    """A live rocrand sampler in the fused loop always holds a pool drawn AHEAD (it rode the optimizer's launch).  set_state
    on it must discard that pool and draw from the new counter (mkb_sampler_set_rng kept the stale pool), and an
    empty batch must move the counter on by one pool."""
    from mkb_amd import datasets, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    ds = datasets.Fb15k237(batch_size=128, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    torch.manual_seed(3)
    m = models.RotatE(hidden_dim=16, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
    ns = sampling.NegativeSampling(size=32, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=9, rng="rocrand")
    opt = optim.Adam([m.entity_embedding, m.relation_embedding], lr=1e-3, lazy_rows=True, draw_ahead=ns, defer_step=True)
    step = FusedTrainStep(m, alpha=1.0)
    w = torch.ones(128, device="cuda")
    s = train[:128]

    def one():
        step.sampled(s, w, ns, "tail-batch")
        opt.step()
        opt.zero_grad()
        return step.negative_sample.clone()

    one()
    state = ns.get_state()        # (a pool is drawn ahead now: the state is the one BEFORE it)
    a = one()
    b = one()
    ns.set_state(*state)          # back to before `a`, with b's successor drawn ahead
    a2 = one()
    b2 = one()
    assert torch.equal(a, a2) and torch.equal(b, b2) and not torch.equal(a, b)
    # plain generate() path as well, and the empty batch
    ns.set_state(*state)
    assert torch.equal(ns.generate(s, "tail-batch"), a)
    before = ns.get_state()
    with pytest.raises(RuntimeError):
        ns.generate(s[:0], "tail-batch")
    after = ns.get_state()
    assert after[1][1] == before[1][1] + 1
    got = ns.generate(s, "tail-batch")  # pool 0 = a, pool 1 went to the empty batch: this is pool 2
    assert torch.equal(got, _third(ns, state, s))
    ns.check()


def _third(ns, state, s):
    """The third pool after `state` (pools 0, 1 = a, b), drawn by a plain generate() sequence."""
    cur = ns.get_state()
    ns.set_state(*state)
    ns.generate(s, "tail-batch")
    ns.generate(s, "tail-batch")
    out = ns.generate(s, "tail-batch")
    ns.set_state(*cur)
    return out


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MKB_FUZZ_SEEDS", "12"))))
def test_random_graph_sampler_vs_c_oracle(liboracle, seed):
    """Random graphs (entity count, relation count, density and hub skew drawn per seed), random K and B: the device sampler
    must equal the plain-C restatement bit for bit -- pools, negatives and the generator state -- over six consecutive batches.
    Dense little graphs drive the table / sort / loop branches of numpy's in1d semantics and big true sets; sparse big ones the
    Bloom / bitmap probes; rows whose true set covers the whole pool are dropped from the batch (the reference hangs on them)."""
    import ctypes

    from mkb_amd import sampling
    from mkb_amd.sampling.negative_sampling import _filter_csr

    rs = np.random.RandomState(9000 + seed)
    N = int(rs.choice([3, 17, 135, 1000, 14541, 60000]))
    R = int(rs.choice([1, 2, 11, 237]))
    T_ = int(min(200000, max(20, N * rs.choice([1, 4, 20]))))
    if rs.rand() < 0.5:  # hub-heavy: Zipf-like entity draws
        pe = 1.0 / np.arange(1, N + 1); pe /= pe.sum()
        h, t = rs.choice(N, size=T_, p=pe), rs.choice(N, size=T_, p=pe)
    else:
        h, t = rs.randint(N, size=T_), rs.randint(N, size=T_)
    train_np = np.unique(np.stack([h, rs.randint(R, size=T_), t], 1).astype(np.int64), axis=0)
    K = int(rs.choice([1, 2, 7, 16, 64, 128, 256, 300, 512]))
    B = int(rs.choice([1, 5, 64, 257, 1024]))
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    triples = [tuple(int(x) for x in row) for row in train_np]
    ns = sampling.NegativeSampling(size=K, train_triples=triples, entities=ents, relations=rels, seed=seed)
    (hk, ho, hv, _), (tk, to, tv, _) = _filter_csr(triples, N, R)
    st = ctypes.create_string_buffer(4 * 624 + 4)
    liboracle.orc_mt_seed(st, ctypes.c_uint32(seed))
    liboracle.orc_generate.restype = ctypes.c_int
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    train = torch.as_tensor(train_np).cuda()
    for c in range(6):
        head = c % 2 == 0
        idx = rs.randint(len(train_np), size=B)
        smp = np.ascontiguousarray(train_np[idx])
        want = np.zeros((B, K), dtype=np.int64)
        pool = np.zeros(2 * K, dtype=np.int64)
        k, o, v, stride = (hk, ho, hv, N) if head else (tk, to, tv, R)
        rc = liboracle.orc_generate(st, ctypes.c_int64(N), ctypes.c_int64(K), p(smp), ctypes.c_int64(B), ctypes.c_int(head),
                                    p(k), ctypes.c_int64(len(k)), p(o), p(v), ctypes.c_int64(stride), p(want), p(pool))
        got = ns.generate(train[torch.as_tensor(idx).cuda()], "head-batch" if head else "tail-batch")
        np.testing.assert_array_equal(got._mkb_pool.pool.cpu().numpy(), pool, err_msg=f"N {N} R {R} K {K} B {B} batch {c}")
        if rc != 0:  # a row whose true set covers the whole pool: the oracle reports it, the device sampler flags it
            with pytest.raises(RuntimeError, match="whole candidate pool"):
                ns.check()
            continue
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg=f"N {N} R {R} K {K} B {B} batch {c}")
    key, pos = ns.get_state()
    np.testing.assert_array_equal(key, np.frombuffer(st.raw[: 4 * 624], dtype=np.uint32))
    assert pos == int(np.frombuffer(st.raw[4 * 624:], dtype=np.int32)[0])

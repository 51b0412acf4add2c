"""-m gpu: pooled (shared candidate pool) path and the fused training step vs the oracle, on real graphs with
negatives from the on-device sampler; Pipeline.learn vs the step-by-step capture of the live reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MODELS = ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"]
ATOL = 1e-4


def _grad_close(got, ref, rtol=0.0, rel=2e-4):
    """Gradients against the oracle: the absolute 1e-5 of the small-shape tests, or `rel` of the reference's largest entry where
    that is tighter -- at the full-size shapes the entries are ~1e-8 .. 1e-6 and an absolute 1e-5 would accept anything."""
    got, ref = np.asarray(got), np.asarray(ref)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=min(1e-5, rel * max(float(np.abs(ref).max()), 1e-30)))


def _setup(cls, name, hidden, B, K, gamma=6.0, seed=0):
    from mkb_amd import datasets, models, sampling
    from oracle import scoring

    ds = getattr(datasets, cls)(batch_size=B, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(seed)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=gamma)
    tb = scoring.Tables(name, hidden, gamma, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone() if hasattr(m, "modulus") else None)
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64))
    return ds, m.cuda(), tb, ns, train


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("mode", ["head-batch", "tail-batch"])
def test_pooled_forward_backward_equals_general_and_oracle(name, mode):
    from mkb_amd import losses
    from oracle import scoring

    ds, m, tb, ns, train = _setup("Umls", name, 50, 77, 16)
    idx = torch.as_tensor(np.random.RandomState(1).randint(len(train), size=77))
    s = train[idx].cuda()
    w = (torch.rand(77) + 0.1).cuda()
    neg = ns.generate(s, mode)
    assert neg._mkb_pool is not None
    plain = neg.clone()  # no pool info -> general kernels
    assert not hasattr(plain, "_mkb_pool")
    ref = scoring.train_step_grads(tb, s.cpu(), neg.cpu(), w.cpu(), mode, 0.5, fast_norm=True)

    got = {}
    for tag, n in (("pooled", neg), ("general", plain)):
        m.zero_grad()
        sc = m(s, n, mode)
        np.testing.assert_allclose(sc.detach().cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        err = losses.Adversarial(alpha=0.5)(m(s), sc, w)
        err.backward()
        np.testing.assert_allclose(err.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        got[tag] = (m.entity_embedding.grad.cpu().numpy().copy(), m.relation_embedding.grad.cpu().numpy().copy())
        _grad_close(got[tag][0], ref["g_ent"].numpy())
        _grad_close(got[tag][1], ref["g_rel"].numpy(), rtol=1e-4)
        if name == "pRotatE":
            np.testing.assert_allclose(m.modulus.grad.cpu().numpy(), ref["g_modulus"].numpy(), rtol=1e-4)
    ns.check()


@pytest.mark.parametrize("cls,name,hidden,B,K", [
    ("Umls", "TransE", 64, 256, 16),          # BASELINE config 1 shape
    ("Wn18rr", "RotatE", 32, 200, 128),       # config 2 sampler shape, reduced dim (oracle in seconds)
    ("Fb15k237", "RotatE", 40, 160, 256),     # headline sampler shape, reduced dim
    ("Fb15k237", "ComplEx", 40, 160, 256),    # config 4
    ("Fb15k237", "DistMult", 33, 96, 256),
    ("Fb15k237", "pRotatE", 24, 96, 256),
    ("Fb15k237", "RotatE", 260, 100, 96),     # forward tile with ragged edges: B not a multiple of 64 (or 8), K not of 64, 260 dims
    ("Wn18rr", "RotatE", 512, 67, 64),        # ... one 64-position tile, 4 units per lane, a single row tile
])
def test_fused_step_vs_oracle_real_graphs(cls, name, hidden, B, K):
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring

    ds, m, tb, ns, train = _setup(cls, name, hidden, B, K, gamma=9.0)
    step = FusedTrainStep(m, alpha=1.0)
    pick = np.random.RandomState(7)
    for c, mode in enumerate(["head-batch", "tail-batch", "head-batch"]):
        idx = torch.as_tensor(pick.randint(len(train), size=B))
        s = train[idx].cuda()
        w = (torch.rand(B) + 0.1).cuda()
        m.zero_grad(set_to_none=True)
        neg = ns.generate(s, mode)
        loss = step(s, w, neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu(), neg.cpu(), w.cpu(), mode, 1.0, fast_norm=True)
        np.testing.assert_allclose(step.positive_score.cpu().numpy(), ref["pos"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(step.negative_score.cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        _grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        _grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
        if name == "pRotatE":
            np.testing.assert_allclose(m.modulus.grad.cpu().numpy(), ref["g_modulus"].numpy(), rtol=1e-4)
    ns.check()


def test_headline_slice_vs_reference_golden(golden):
    """FB15k-237 RotatE hidden=1000 K=256: 16 rows of a real batch vs scores, loss AND gradients captured from the live
    reference (tools/make_golden.py::gen_headline_slice).  The tables are drawn with numpy's legacy generator (see
    util_gpu.headline_tables), so nothing here depends on the torch CPU RNG stream of the build container."""
    from mkb_amd import _hip, datasets, losses, models
    from mkb_amd.fused import FusedTrainStep
    from mkb_amd.sampling.negative_sampling import PoolInfo
    from util_gpu import headline_tables

    g = golden("headline_slice.npz")
    ds = datasets.Fb15k237(batch_size=16, shuffle=False, seed=42, num_workers=0)
    m = models.RotatE(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9)
    ent, rel = headline_tables(seed=int(g["table_seed"]))
    with torch.no_grad():
        m.entity_embedding.copy_(torch.from_numpy(ent))
        m.relation_embedding.copy_(torch.from_numpy(rel))
    m = m.cuda()
    s = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)[g["idx"]]).cuda()
    w = torch.as_tensor(g["weight"]).cuda()

    def check_grads(mode, tag):
        ge = m.entity_embedding.grad.cpu().numpy()
        gr = m.relation_embedding.grad.cpu().numpy()
        rows = g[f"{mode}/g_ent_rows"].astype(np.int64)
        other = np.ones(ge.shape[0], dtype=bool)
        other[rows] = False
        assert not ge[other].any(), f"{tag}: gradient outside the rows the reference touched"
        np.testing.assert_allclose(ge[rows][:, ::8], g[f"{mode}/g_ent_cols8"], rtol=0, atol=1e-5, err_msg=tag)
        np.testing.assert_allclose(ge[rows].astype(np.float64).sum(1), g[f"{mode}/g_ent_rowsum"], rtol=0, atol=2e-4, err_msg=tag)
        np.testing.assert_allclose((ge[rows].astype(np.float64) ** 2).sum(1), g[f"{mode}/g_ent_rowsq"], rtol=1e-3, atol=1e-9,
                                   err_msg=tag)
        rrows = g[f"{mode}/g_rel_rows"].astype(np.int64)
        other = np.ones(gr.shape[0], dtype=bool)
        other[rrows] = False
        assert not gr[other].any()
        np.testing.assert_allclose(gr[rrows], g[f"{mode}/g_rel"], rtol=1e-4, atol=1e-5, err_msg=tag)

    for mode in ["head-batch", "tail-batch"]:
        neg = torch.as_tensor(g[f"{mode}/neg"].astype(np.int64)).cuda()
        # (a) the autograd route over the general kernels (no pool description on these negatives; 16 x 256 slots is
        #     below the auto-discovery threshold)
        m.zero_grad(set_to_none=True)
        pos_score, neg_score = m(s), m(s, neg, mode)
        np.testing.assert_allclose(pos_score.detach().cpu().numpy(), g[f"{mode}/pos"], rtol=0, atol=ATOL)
        np.testing.assert_allclose(neg_score.detach().cpu().numpy(), g[f"{mode}/score"], rtol=0, atol=ATOL)
        err = losses.Adversarial(alpha=float(g["alpha"]))(pos_score, neg_score, w)
        err.backward()
        np.testing.assert_allclose(err.item(), float(g[f"{mode}/loss"]), rtol=0, atol=1e-5)
        check_grads(mode, f"general {mode}")
        # (b) the fused pooled step on the same negatives (their shared pool recovered by PoolInfo.discover)
        info = PoolInfo.discover(neg, s, _hip.mode_id(mode))
        assert info is not None
        neg._mkb_pool = info
        m.zero_grad(set_to_none=True)
        step = FusedTrainStep(m, alpha=float(g["alpha"]))
        loss = step(s, w, neg, mode)
        np.testing.assert_allclose(step.negative_score.cpu().numpy(), g[f"{mode}/score"], rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), float(g[f"{mode}/loss"]), rtol=0, atol=1e-5)
        check_grads(mode, f"fused {mode}")


@pytest.mark.parametrize("name", ["RotatE", "ComplEx", "TransE"])
def test_full_size_fused_step_gradients_vs_oracle_on_a_128_row_slice(name):
    """BASELINE full size (FB15k-237, hidden 1000, K=256, B=1024), loss and BOTH dense gradients against the oracle.
    A full oracle step costs ~80 s (RotatE), so 128 rows carry the weight and the other 896 get weight 0: their loss
    terms and gradient seeds are exactly 0 (adversarial.py:21-30: every term is multiplied by w_i / W), hence the
    1024-row launch -- same grid, same tiles, same pool -- must reproduce the oracle's step over the 128 rows alone.
    The slice mixes full row tiles (rows 0-63) with isolated rows (every 14th afterwards)."""
    from mkb_amd import datasets, models, sampling
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring

    B, K, hidden = 1024, 256, 1000
    ds = datasets.Fb15k237(batch_size=B, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(11)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=9.0)
    tb = scoring.Tables(name, hidden, 9.0, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone() if hasattr(m, "modulus") else None)
    m = m.cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64))
    idx = torch.as_tensor(np.random.RandomState(9).randint(len(train), size=B))
    s = train[idx].cuda()
    rows = torch.cat([torch.arange(64), torch.arange(64, B, 14)[:64]])
    assert len(rows) == 128
    w = torch.zeros(B)
    w[rows] = torch.rand(128) + 0.1
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        m.zero_grad(set_to_none=True)
        step = FusedTrainStep(m, alpha=1.0)
        loss = step(s, w.cuda(), neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu()[rows], neg.cpu()[rows], w[rows], mode, 1.0, fast_norm=True)
        np.testing.assert_allclose(step.positive_score.cpu()[rows].numpy(), ref["pos"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(step.negative_score.cpu()[rows].numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        _grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        _grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
    ns.check()


EVERY_ROW_CASES = [("RotatE", "head-batch"), ("RotatE", "tail-batch"), ("ComplEx", "head-batch"), ("ComplEx", "tail-batch"),
                   ("DistMult", "head-batch"), ("DistMult", "tail-batch"), ("TransE", "head-batch"), ("TransE", "tail-batch"),
                   ("pRotatE", "tail-batch")]


@pytest.mark.parametrize("name,mode", EVERY_ROW_CASES)
def test_headline_full_size_every_row_weighted_vs_oracle(name, mode):
    """The full-size launch (FB15k-237, hidden 1000, K 256, B 1024) with ALL 1024 rows carrying weight (the slice tests above
    weight 128): loss, every score and BOTH dense gradients against the oracle, for every model family and both corruption
    modes (rotate.py:83-97, complex.py:65-85, distmult.py:63-75, transe.py:65-76 -- ComplEx / DistMult run on the matrix
    route, whose row clamp once hid in the last three batch rows).  The oracle walks the batch in 8 chunks of 128 rows
    (memory: [128, 256, 2000] per chunk) -- the loss is a weighted sum over rows with the global normaliser W
    (adversarial.py:28-29), so the chunks' results, each rescaled by W_chunk / W, add up to the step of the whole batch.
    RotatE's oracle runs with ``fast_norm=True`` here: sqrt(re^2 + im^2) instead of the reference's stack -> norm(dim=0)
    (rotate.py:95-96; a torch-CPU pathology, ~50x slower) -- ~1 ulp per term apart, far inside the 1e-4 of this test; the
    reference-exact form is held on the 128-row tests/golden/headline_slice.npz (test_headline_slice_vs_reference_golden) and
    the two oracle forms against each other in tests/test_oracle_scoring.py."""
    from mkb_amd import datasets, models, sampling
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring

    B, K, hidden = 1024, 256, 1000
    ds = datasets.Fb15k237(batch_size=B, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(12)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=9.0)
    tb = scoring.Tables(name, hidden, 9.0, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone() if hasattr(m, "modulus") else None)
    m = m.cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64))
    s = train[torch.as_tensor(np.random.RandomState(10).randint(len(train), size=B))].cuda()
    w = torch.rand(B) + 0.1
    neg = ns.generate(s, mode)
    ns.check()
    step = FusedTrainStep(m, alpha=1.0)
    loss = step(s, w.cuda(), neg, mode)
    got_pos, got_neg = step.positive_score.cpu(), step.negative_score.cpu()
    g_ent, g_rel, total = torch.zeros_like(tb.ent), torch.zeros_like(tb.rel), 0.0
    W = w.sum()
    for lo in range(0, B, 128):
        rows = slice(lo, lo + 128)
        ref = scoring.train_step_grads(tb, s.cpu()[rows], neg.cpu()[rows], w[rows], mode, 1.0, fast_norm=True)
        scale = w[rows].sum() / W
        g_ent += ref["g_ent"] * scale
        g_rel += ref["g_rel"] * scale
        total += float(ref["loss"]) * float(scale)
        np.testing.assert_allclose(got_pos[rows].numpy(), ref["pos"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(got_neg[rows].numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
    np.testing.assert_allclose(loss.item(), total, rtol=0, atol=1e-5)
    _grad_close(m.entity_embedding.grad.cpu().numpy(), g_ent.numpy())
    _grad_close(m.relation_embedding.grad.cpu().numpy(), g_rel.numpy(), rtol=1e-4)


def _weighted_slice_step(cls, name, hidden, B, K, n_rows, gamma, alpha, seed=11):
    """Fused step over B rows of which only `n_rows` carry weight, against the oracle's step over those rows alone (the
    argument of test_full_size_fused_step_gradients_vs_oracle_on_a_128_row_slice)."""
    from mkb_amd import datasets, models, sampling
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring

    ds = getattr(datasets, cls)(batch_size=B, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(seed)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=gamma)
    tb = scoring.Tables(name, hidden, gamma, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone() if hasattr(m, "modulus") else None)
    m = m.cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64))
    idx = torch.as_tensor(np.random.RandomState(9).randint(len(train), size=B))
    s = train[idx].cuda()
    half = n_rows // 2
    rows = torch.cat([torch.arange(half), torch.arange(half, B, (B - half) // half)[:half]])
    assert len(rows) == n_rows
    w = torch.zeros(B)
    w[rows] = torch.rand(n_rows) + 0.1
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        m.zero_grad(set_to_none=True)
        step = FusedTrainStep(m, alpha=alpha)
        loss = step(s, w.cuda(), neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu()[rows], neg.cpu()[rows], w[rows], mode, alpha, fast_norm=True)
        np.testing.assert_allclose(step.positive_score.cpu()[rows].numpy(), ref["pos"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(step.negative_score.cpu()[rows].numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        _grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        _grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
        if name == "pRotatE":
            np.testing.assert_allclose(m.modulus.grad.cpu().numpy(), ref["g_modulus"].numpy(), rtol=1e-4)
    ns.check()


def test_config2_full_size_fused_step_vs_oracle():
    """BASELINE configs[1] at FULL size: WN18RR, RotatE hidden 500, K = 128, B = 1024 (reference shape:
    mkb/datasets/wn18rr.py:44-47), loss and both dense gradients against the oracle on a 128-row slice."""
    _weighted_slice_step("Wn18rr", "RotatE", 500, 1024, 128, 128, gamma=6.0, alpha=0.5)


def test_config2_full_size_pooled_equals_general():
    """configs[1] at full size, all 1024 rows weighted: the pooled route (shared-pool kernels) and the general route
    (arbitrary-candidate kernels + autograd) must agree on scores, loss and both gradients."""
    import mkb_amd.models.base as mb
    from mkb_amd import losses

    B, K = 1024, 128
    ds, m, tb, ns, train = _setup("Wn18rr", "RotatE", 500, B, K, gamma=6.0)
    idx = torch.as_tensor(np.random.RandomState(3).randint(len(train), size=B))
    s = train[idx].cuda()
    w = (torch.rand(B) + 0.1).cuda()
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        plain = neg.clone()
        got = {}
        for tag, n in (("pooled", neg), ("general", plain)):
            m.zero_grad(set_to_none=True)
            mb.AUTO_POOL = tag == "pooled"  # (131 k slots without a pool description would be searched for their pool, and found)
            try:
                sc = m(s, n, mode)
            finally:
                mb.AUTO_POOL = True
            err = losses.Adversarial(alpha=0.5)(m(s), sc, w)
            err.backward()
            got[tag] = (sc.detach().cpu().numpy(), err.item(), m.entity_embedding.grad.cpu().numpy().copy(),
                        m.relation_embedding.grad.cpu().numpy().copy())
        np.testing.assert_allclose(got["pooled"][0], got["general"][0], rtol=0, atol=ATOL)
        np.testing.assert_allclose(got["pooled"][1], got["general"][1], rtol=0, atol=1e-5)
        for k in (2, 3):  # relative to the gradients' scale (an absolute 1e-5 is larger than most entries at this size)
            np.testing.assert_allclose(got["pooled"][k], got["general"][k], rtol=0, atol=1e-5 * np.abs(got["general"][k]).max())
    ns.check()


@pytest.mark.parametrize("name", MODELS)
def test_headline_shape_pooled_equals_general_on_every_row(name):
    """FB15k-237, hidden 1000, K 256, B 1024, all rows weighted: the pooled route and the general route (arbitrary-candidate
    kernels + autograd: an independent implementation of the same functions) on EVERY row of both gradients, to a tolerance
    RELATIVE to the gradients' scale -- at this size they are ~1e-7 and an absolute 1e-5 sees nothing (a row clamp in the
    128-row GEMM tile kernel went unnoticed for two rounds that way: tests/test_gpu_gemm_bf16x3.py)."""
    import mkb_amd.models.base as mb
    from mkb_amd import losses

    B, K = 1024, 256
    ds, m, tb, ns, train = _setup("Fb15k237", name, 1000, B, K, gamma=9.0)
    idx = torch.as_tensor(np.random.RandomState(11).randint(len(train), size=B))
    s = train[idx].cuda()
    w = (torch.rand(B) + 0.1).cuda()
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        plain = neg.clone()
        got = {}
        assert getattr(neg, "_mkb_pool", None) is not None and getattr(plain, "_mkb_pool", None) is None
        for tag, n in (("pooled", neg), ("general", plain)):
            m.zero_grad(set_to_none=True)
            mb.AUTO_POOL = tag == "pooled"  # (negatives without a pool description would otherwise be searched for their pool)
            try:
                sc = m(s, n, mode)
            finally:
                mb.AUTO_POOL = True
            err = losses.Adversarial(alpha=1.0)(m(s), sc, w)
            err.backward()
            got[tag] = (sc.detach().cpu().numpy(), err.item(), m.entity_embedding.grad.cpu().numpy().copy(),
                        m.relation_embedding.grad.cpu().numpy().copy())
        np.testing.assert_allclose(got["pooled"][0], got["general"][0], rtol=0, atol=ATOL)
        np.testing.assert_allclose(got["pooled"][1], got["general"][1], rtol=0, atol=1e-5)
        for k in (2, 3):
            scale = np.abs(got["general"][k]).max()
            assert scale > 0
            np.testing.assert_allclose(got["pooled"][k], got["general"][k], rtol=0, atol=1e-5 * scale)
    ns.check()


@pytest.mark.parametrize("name,hidden,B,K", [
    ("TransE", 500, 2048, 384),    # 768 positions / 4 blocks = 192 per block = 3 halves of 64 (rounded up to 4)
    ("pRotatE", 500, 2048, 384),
    ("TransE", 201, 4096, 300),    # odd rows (one unit per lane, 8-half accumulator): 600 positions / 2 blocks = 5 halves -> 8
])
def test_single_pass_backward_position_blocks_not_a_power_of_two(name, hidden, B, K):
    """Shapes whose position blocks hold 3, 5, 6 or 7 sixty-four-slot halves: the host rounds the halves up to a power of
    two (kernel chunking, seed layout and workspace sizes all assume one); the gradients must still match the oracle."""
    _weighted_slice_step("Fb15k237", name, hidden, B, K, 64, gamma=9.0, alpha=1.0)


@pytest.mark.parametrize("fuse", [True, False])
def test_pipeline_countries_vs_reference_capture(golden, fuse, capsys):
    """compose/pipeline.py:79-129 setup: CountriesS1 bs=20 seed 42, RotatE hidden 5 gamma 3, K=4, Adam 5e-5,
    alpha .5, 3 epochs, eval every epoch.  Every step's (sample, negatives, loss) and the final tables / metrics
    must reproduce the capture of the live reference."""
    from mkb_amd import compose, datasets, evaluation, losses, models, sampling

    g = golden("pipeline_countries.npz")
    gj = golden("pipeline_countries.json")
    torch.manual_seed(42)
    ds = datasets.CountriesS1(batch_size=20, seed=42)
    model = models.RotatE(hidden_dim=5, entities=ds.entities, relations=ds.relations, gamma=3)
    np.testing.assert_array_equal(model.entity_embedding.detach().numpy(), g["ent0"])
    model = model.cuda()
    ns = sampling.NegativeSampling(size=4, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=0.00005)
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=8,
                               device="cuda")
    rec = {"sample": [], "neg": [], "loss": []}
    gen0 = ns.generate

    def gen(sample, mode):
        n = gen0(sample=sample, mode=mode)
        rec["sample"].append(sample.cpu().numpy())
        rec["neg"].append(n.cpu().numpy())
        return n

    ns.generate = gen
    pipe = compose.Pipeline(epochs=3, eval_every=1, early_stopping_rounds=3, device="cuda")
    pipe.fuse = fuse
    upd0 = pipe.metric_loss.update
    pipe.metric_loss.update = lambda x: (rec["loss"].append(x), upd0(x))[1]
    pipe = pipe.learn(model=model, dataset=ds, evaluation=ev, sampling=ns, optimizer=opt, loss=losses.Adversarial(alpha=0.5))
    assert len(rec["sample"]) == int(g["steps"])
    for i, (s, n) in enumerate(zip(rec["sample"], rec["neg"])):
        np.testing.assert_array_equal(s, g[f"step{i}/sample"])
        np.testing.assert_array_equal(n, g[f"step{i}/neg"])
    np.testing.assert_allclose(rec["loss"], g["loss"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(model.entity_embedding.detach().cpu().numpy(), g["ent_final"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(model.relation_embedding.detach().cpu().numpy(), g["rel_final"], rtol=0, atol=2e-5)
    for split in ("valid_scores", "test_scores"):
        for k, v in gj[split].items():
            assert abs(getattr(pipe, split)[k] - v) <= (0.5 if k.startswith("MR") and not k.startswith("MRR") else 0.02), (split, k)


def test_evaluation_doctest_known_answer(golden):
    """evaluation/evaluation.py:43-119: trained toy RotatE -> {'MRR': 0.5417, 'MR': 2.25, 'HITS@1': 0.25, ...}."""
    from mkb_amd import evaluation, models

    g = golden("evaluation.npz")
    gj = golden("evaluation.json")
    ents = {f"e{i}": i for i in range(4)}
    rels = {"r0": 0, "r1": 1}
    m = models.RotatE(hidden_dim=3, entities=ents, relations=rels, gamma=1)
    m._set_params(torch.as_tensor(g["ent"]), torch.as_tensor(g["rel"]))
    m = m.cuda().eval()
    train = [(0, 0, 1), (0, 1, 1), (2, 0, 3), (2, 1, 3)]
    test = [(0, 0, 1), (2, 1, 3)]
    ev = evaluation.Evaluation(true_triples=train + test + test, entities=ents, relations=rels, batch_size=2, device="cuda")
    assert ev.eval(model=m, dataset=test) == gj["toy_eval"] == {"MRR": 0.5417, "MR": 2.25, "HITS@1": 0.25, "HITS@3": 1.0, "HITS@10": 1.0}
    assert ev.eval_relations(model=m, dataset=test) == gj["toy_eval_relations"]


@pytest.mark.parametrize("name", ["TransE", "ComplEx", "RotatE"])
def test_evaluation_countries_vs_reference(golden, name):
    from mkb_amd import datasets, evaluation, models

    g = golden("evaluation.npz")
    gj = golden("evaluation.json")
    ds = datasets.CountriesS1(batch_size=20, seed=42)
    m = getattr(models, name)(hidden_dim=6, entities=ds.entities, relations=ds.relations, gamma=4)
    m._set_params(torch.as_tensor(g[f"countries/{name}/ent"]), torch.as_tensor(g[f"countries/{name}/rel"]))
    m = m.cuda().eval()
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=8,
                               device="cuda")
    for split in ("test", "valid"):
        got = ev.eval(model=m, dataset=getattr(ds, split))
        want = gj[f"countries/{name}/{split}"]
        for k in want:
            assert abs(got[k] - want[k]) <= (0.1 if k == "MR" else 2e-3), (split, k, got, want)


def test_sharded_batch_with_global_weight_sum_equals_full_batch():
    """Data-parallel math on one GPU: two half-batches stepped with the GLOBAL normaliser W (weight_sum=) leave
    the same dense gradients and total loss as one step over the whole batch (the sampler state is replayed so
    both runs see the identical pool)."""
    from mkb_amd.fused import FusedTrainStep

    ds, m, tb, ns, train = _setup("Fb15k237", "RotatE", 32, 128, 64, gamma=9.0)
    idx = torch.as_tensor(np.random.RandomState(3).randint(len(train), size=128))
    s = train[idx].cuda()
    w = (torch.rand(128) + 0.1).cuda()
    step = FusedTrainStep(m, alpha=1.0)
    ns.generate(s[:4], "tail-batch")                 # create the device handle
    key, pos = ns.get_state()
    neg = ns.generate(s, "tail-batch")
    m.zero_grad(set_to_none=True)
    full_loss = step(s, w, neg, "tail-batch").item()
    g_full = (m.entity_embedding.grad.clone(), m.relation_embedding.grad.clone())
    m.zero_grad(set_to_none=True)
    W = w.sum().reshape(1)
    total = 0.0
    for lo, hi in ((0, 64), (64, 128)):              # "rank 0" and "rank 1": same RNG state -> same pool
        ns.set_state(key, pos)
        negh = ns.generate(s[lo:hi].contiguous(), "tail-batch")
        np.testing.assert_array_equal(negh.cpu().numpy(), neg[lo:hi].cpu().numpy())
        total += step(s[lo:hi].contiguous(), w[lo:hi].contiguous(), negh, "tail-batch", weight_sum=W).item()
    np.testing.assert_allclose(total, full_loss, rtol=0, atol=1e-6)
    np.testing.assert_allclose(m.entity_embedding.grad.cpu().numpy(), g_full[0].cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(m.relation_embedding.grad.cpu().numpy(), g_full[1].cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("hidden", [37, 64])
@pytest.mark.parametrize("name", ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"])
def test_device_ranking_equals_reference_route(name, hidden):
    """mkb_rank (tiled all-entity forward + on-device filtered count) == TestDataset + general forward + argsort.
    (hidden 64: RotatE and TransE take the pooled forward's register tile for the all-entity block, the others and hidden 37 the
    lane-owns-dims kernel.)"""
    from mkb_amd import datasets, evaluation, models

    ds = datasets.Umls(batch_size=8, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(3)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=6).cuda().eval()
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=64,
                               device="cuda", num_workers=0)
    test = ds.test[:150]
    fast = ev.eval(model=m, dataset=test)
    fast_rel = ev.eval_relations(model=m, dataset=test)
    ev.force_reference_path = True
    slow = ev.eval(model=m, dataset=test)
    slow_rel = ev.eval_relations(model=m, dataset=test)
    assert fast == slow, (fast, slow)
    assert fast_rel == slow_rel, (fast_rel, slow_rel)  # relation ranking: device searchsorted filter vs TestDatasetRelation


@pytest.mark.parametrize("name", ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"])
def test_device_ranking_at_the_headline_size_equals_reference_route(name):
    """The same comparison on FB15k-237 with hidden 1000 (14,541 candidates per query, the tiled all-entity forward at its
    full row length): 96 test triples, both modes.  At this size neighbouring candidates' scores are ~1e-5 apart and the two
    routes sum 1000-2000 terms in different orders, so single ranks may swap: the means must agree closely, which a wrong
    tile or a mis-addressed block of entities (hundreds of ranks per query) would not survive."""
    from mkb_amd import datasets, evaluation, models

    ds = datasets.Fb15k237(batch_size=8, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(5)
    m = getattr(models, name)(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9).cuda().eval()
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=32,
                               device="cuda", num_workers=0)
    test = ds.test[:96]
    fast = ev.eval(model=m, dataset=test)
    ev.force_reference_path = True
    slow = ev.eval(model=m, dataset=test)
    assert abs(fast["MR"] - slow["MR"]) <= 0.002 * slow["MR"] + 0.5, (fast, slow)
    assert abs(fast["MRR"] - slow["MRR"]) <= 2e-3, (fast, slow)
    for k in ("HITS@1", "HITS@3", "HITS@10"):
        assert abs(fast[k] - slow[k]) <= 0.011, (fast, slow)


@pytest.mark.parametrize("name,hidden,world,table", [("RotatE", 48, 2, "small"), ("ComplEx", 32, 4, "small"),
                                                      ("TransE", 500, 2, "small"), ("pRotatE", 40, 2, "small"),
                                                      ("DistMult", 37, 3, "small"), ("RotatE", 24, 2, "big"),
                                                      ("pRotatE", 20, 3, "big")])
def test_dim_sharded_training_equals_single_device(name, hidden, world, table):
    """Embedding-dimension sharding over `world` processes (gloo, all on this GPU): 6 fused steps + lazy Adam leave
    the same tables and losses as the single-process fused step.  "big": FB15k-237's 14,541 entities, so the table steps
    row-lazily, with the sampler riding the optimizer launch and the real step deferred on the sharded side."""
    import os
    import socket
    import subprocess
    import sys

    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "tp_worker.py"), name, str(hidden), "16", table]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert out.returncode == 0 and "TP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("name,hidden,world,size", [("RotatE", 24, 2, "small"), ("TransE", 33, 3, "small"), ("ComplEx", 16, 4, "small"),
                                                    ("pRotatE", 20, 2, "small"), ("RotatE", 40, 2, "big"), ("TransE", 32, 3, "big"),
                                                    ("RotatE", 40, 1, "big"),  # world 1: no collective runs, rows move without copies
                                                    ("RotatE", 24, 2, "wide")])  # 2200 rows per rank: the routes' requests are NOT merged
def test_row_sharded_table_training_equals_single_device(name, hidden, world, size):
    """BASELINE config 5's partitioning (mkb_amd.table_rows): entity rows, their gradient and their Adam state sharded by
    row over `world` processes (gloo, all on this one GPU), the fused HIP step running on the compact table of each rank.
    Losses and reassembled tables must equal the single-process run (tests/tr_worker.py)."""
    import os
    import socket
    import subprocess
    import sys

    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "tr_worker.py"), name, str(hidden), "16", size]
    # "big": FB15k-237 -- shards of > 4096 rows, so each shard steps row-lazily with the real step deferred
    # (mkb_adam_rows_advance_sharded), against the single-process run with the same optimizer settings
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert out.returncode == 0 and "TR_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("name", ["RotatE", "TransE", "ComplEx"])
@pytest.mark.parametrize("B,K,hidden", [(1, 3, 6), (13, 5, 33), (9, 7, 130)])
def test_fused_step_edge_shapes_with_wrapping_rows(name, B, K, hidden):
    """Tiny entity set: the filter removes most of the pool, rows wrap around their survivors (a pool position is
    used several times by one row: multiplicities > 1), B is not a multiple of the row tile, odd dims (scalar-lane
    kernels) and even dims (vector-lane kernels)."""
    from mkb_amd import models, sampling
    from mkb_amd.fused import FusedTrainStep, pooled_supported
    from oracle import scoring

    N, R = 6, 2
    ents, rels = {i: i for i in range(N)}, {i: i for i in range(R)}
    rs = np.random.RandomState(B * 31 + K)
    train = sorted({(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(60)})  # dense: big true sets
    torch.manual_seed(B + K)
    m = getattr(models, name)(hidden_dim=hidden, entities=ents, relations=rels, gamma=4.0)
    tb = scoring.Tables(name, hidden, 4.0, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone() if hasattr(m, "modulus") else None)
    m = m.cuda()
    assert pooled_supported(m, B, K)
    ns = sampling.NegativeSampling(size=K, train_triples=train, entities=ents, relations=rels, seed=11)
    step = FusedTrainStep(m, alpha=0.5)
    wrapped = False
    for it, mode in enumerate(["tail-batch", "head-batch"] * 8):
        idx = rs.randint(len(train), size=B)
        s = torch.tensor([train[i] for i in idx]).cuda()
        w = (torch.rand(B) + 0.1).cuda()
        try:
            neg = ns.generate(s, mode)
            ns.check()
        except RuntimeError:  # a row's true set covers the whole pool: the reference would hang; skip that batch
            ns = sampling.NegativeSampling(size=K, train_triples=train, entities=ents, relations=rels, seed=100 + it)
            continue
        wrapped |= bool((neg._mkb_pool.cnt.to(torch.int32) > 1).any())
        m.zero_grad(set_to_none=True)
        loss = step(s, w, neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu(), neg.cpu(), w.cpu(), mode, 0.5, fast_norm=True)
        np.testing.assert_allclose(step.negative_score.cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        _grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        _grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
    assert wrapped, "the test is meant to exercise multiplicities > 1"


@pytest.mark.parametrize("name,hidden,B,K", [("TransE", 20, 8, 600), ("RotatE", 40, 70, 600), ("ComplEx", 24, 64, 700),
                                             ("pRotatE", 16, 33, 1024), ("RotatE", 300, 64, 640)])
def test_large_pools_run_on_the_pooled_kernels(name, hidden, B, K):
    """512 < size <= 1024 (pools of up to 2048 positions, the device sampler's limit): the pooled kernels cover them since
    round 3 (two-pass backward where the single-pass accumulator does not fit, the loss rows re-reading long rows).  Scores
    through model(...), then the fused step's loss and gradients, against the oracle; negatives against the oracle sampler."""
    from mkb_amd.fused import FusedTrainStep, pooled_supported
    from oracle import sampler as osamp
    from oracle import scoring

    ds, m, tb, ns, train = _setup("Fb15k237", name, hidden, B, K, gamma=9.0)
    assert pooled_supported(m, B, K)
    idx = torch.as_tensor(np.random.RandomState(5).randint(len(train), size=B))
    s = train[idx].cuda()
    w = (torch.rand(B) + 0.1).cuda()
    on = osamp.NegativeSampling(K, ds.train, ds.entities, ds.relations, seed=42)
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        want, _ = on.generate(s.cpu().numpy(), mode)
        np.testing.assert_array_equal(neg.cpu().numpy(), want)
        assert neg._mkb_pool.usable_for(m, s, neg._mkb_pool.mode_id)
        got = m(s, neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu(), neg.cpu(), w.cpu(), mode, 1.0, fast_norm=True)
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        m.zero_grad(set_to_none=True)
        step = FusedTrainStep(m, alpha=1.0)
        loss = step(s, w, neg, mode)
        np.testing.assert_allclose(step.negative_score.cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        _grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        _grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
    ns.check()


@pytest.mark.parametrize("name,hidden", [("RotatE", 1000), ("ComplEx", 1000), ("TransE", 1000)])
def test_full_size_pooled_path_agrees_with_general_kernels_and_oracle_rows(name, hidden, monkeypatch):
    """BASELINE full size (FB15k-237, hidden 1000, K=256, B=1024): the oracle needs ~80 s per step here, so
    (a) the fused pooled step is checked against the GENERAL kernels (an independent implementation: per-row LDS
        query + wave-per-candidate forward, atomics backward) on every score, the loss and both dense gradients;
    (b) the oracle itself checks a 12-row slice of the same batch (scores), and the loss is recomputed from the
        step's own scores with torch ops (losses/adversarial.py:21-30 formula)."""
    from mkb_amd import datasets, losses, models, sampling
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring
    import mkb_amd.models.base as model_base

    monkeypatch.setattr(model_base, "AUTO_POOL", False)  # the plain copy below must really take the general kernels
    ds = datasets.Fb15k237(batch_size=1024, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(42)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=9.0)
    tb = scoring.Tables(name, hidden, 9.0, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone() if hasattr(m, "modulus") else None)
    m = m.cuda()
    ns = sampling.NegativeSampling(size=256, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64))
    idx = torch.as_tensor(np.random.RandomState(5).randint(len(train), size=1024))
    s, w = train[idx].cuda(), (torch.rand(1024) + 0.1).cuda()
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        m.zero_grad(set_to_none=True)
        step = FusedTrainStep(m, alpha=1.0)
        loss = step(s, w, neg, mode)
        pos_f, neg_f = step.positive_score.clone(), step.negative_score.clone()
        g_f = (m.entity_embedding.grad.clone(), m.relation_embedding.grad.clone())
        # (a) general kernels through autograd on a plain copy of the negatives
        m.zero_grad(set_to_none=True)
        plain = neg.clone()
        pos_g, neg_g = m(s), m(s, plain, mode)
        err = losses.Adversarial(alpha=1.0)(pos_g, neg_g, w)
        err.backward()
        np.testing.assert_allclose(pos_f.cpu().numpy(), pos_g.detach().cpu().numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(neg_f.cpu().numpy(), neg_g.detach().cpu().numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), err.item(), rtol=0, atol=1e-5)
        _grad_close(g_f[0].cpu().numpy(), m.entity_embedding.grad.cpu().numpy())
        _grad_close(g_f[1].cpu().numpy(), m.relation_embedding.grad.cpu().numpy(), rtol=1e-4)
        # (b) oracle on a slice + the loss formula on the step's own scores
        rows = torch.arange(0, 1024, 93)[:12]
        ref = scoring.score(tb, s[rows.cuda()].cpu(), neg[rows.cuda()].cpu(), mode, fast_norm=True)
        np.testing.assert_allclose(neg_f[rows.cuda()].cpu().numpy(), ref.numpy(), rtol=0, atol=ATOL)
        ref_loss = scoring.adversarial(pos_f.cpu(), neg_f.cpu(), w.cpu(), 1.0)
        np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=0, atol=1e-5)
    ns.check()


@pytest.mark.parametrize("name,hidden", [("RotatE", 1500), ("RotatE", 3000), ("pRotatE", 1500), ("pRotatE", 3000),
                                          ("TransE", 3000), ("TransE", 301), ("DistMult", 2600), ("RotatE", 257),
                                          ("ComplEx", 700)])
def test_every_launch_configuration_agrees_with_general_kernels(name, hidden, monkeypatch):
    """The pooled kernels are instantiated per (units per lane, waves per workgroup); wide rows use 16-wave
    workgroups whose register budget is tight (some instantiations spill to scratch).  Every configuration of the
    launch table must give the general kernels' scores and gradients (MFMA route off so the tile kernels run)."""
    from mkb_amd import losses
    monkeypatch.setenv("MKB_POOL_NO_MFMA", "1")
    ds, m, tb, ns, train = _setup("Umls", name, hidden, 40, 16)
    idx = torch.as_tensor(np.random.RandomState(3).randint(len(train), size=40))
    s, w = train[idx].cuda(), (torch.rand(40) + 0.1).cuda()
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        plain = neg.clone()
        got = {}
        for tag, n in (("pooled", neg), ("general", plain)):
            m.zero_grad(set_to_none=True)
            sc = m(s, n, mode)
            err = losses.Adversarial(alpha=0.5)(m(s), sc, w)
            err.backward()
            got[tag] = (sc.detach().cpu().numpy(), err.item(), m.entity_embedding.grad.cpu().numpy().copy(),
                        m.relation_embedding.grad.cpu().numpy().copy())
        scale = max(1.0, float(np.abs(got["general"][0]).max()))
        np.testing.assert_allclose(got["pooled"][0], got["general"][0], rtol=0, atol=ATOL * scale)
        np.testing.assert_allclose(got["pooled"][1], got["general"][1], rtol=0, atol=1e-5 * scale)
        _grad_close(got["pooled"][2], got["general"][2])
        _grad_close(got["pooled"][3], got["general"][3], rtol=1e-4)
    ns.check()


@pytest.mark.parametrize("name", ["RotatE", "ComplEx"])
def test_foreign_negatives_are_scanned_for_their_shared_pool(name, monkeypatch):
    """Negatives that do not come from mkb_amd's sampler (here: the oracle's restatement of the reference sampler, made
    on the host) carry no pool description; a big enough batch is scanned for one (PoolInfo.discover) and then scored
    by the pooled kernels.  Scores and gradients must equal the general kernels'."""
    from mkb_amd import datasets, losses, models
    from mkb_amd.sampling.negative_sampling import PoolInfo
    import mkb_amd.models.base as model_base
    from oracle import sampler as oracle_sampler

    ds = datasets.Fb15k237(batch_size=512, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(1)
    m = getattr(models, name)(hidden_dim=96, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
    on = oracle_sampler.NegativeSampling(size=64, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=5)
    train = np.asarray(ds.train, dtype=np.int64)
    s_np = train[np.random.RandomState(2).randint(len(train), size=512)]
    neg_np, _ = on.generate(s_np, "head-batch")
    s, neg, w = torch.as_tensor(s_np).cuda(), torch.as_tensor(neg_np).cuda(), (torch.rand(512) + 0.1).cuda()
    found = []
    real = PoolInfo.discover.__func__
    monkeypatch.setattr(PoolInfo, "discover", classmethod(lambda cls, *a: found.append(real(cls, *a)) or found[-1]))
    got = {}
    for auto in (True, False):
        monkeypatch.setattr(model_base, "AUTO_POOL", auto)
        m.zero_grad(set_to_none=True)
        sc = m(s, neg, "head-batch")
        losses.Adversarial(alpha=1.0)(m(s), sc, w).backward()
        got[auto] = (sc.detach().cpu().numpy(), m.entity_embedding.grad.cpu().numpy().copy(),
                     m.relation_embedding.grad.cpu().numpy().copy())
    assert len(found) == 1 and found[0] is not None and found[0].pool.numel() == 128  # 512 x 64 slots: scanned once
    np.testing.assert_allclose(got[True][0], got[False][0], rtol=0, atol=ATOL)
    _grad_close(got[True][1], got[False][1])
    _grad_close(got[True][2], got[False][2], rtol=1e-4)


def test_diverse_foreign_negatives_stay_on_the_general_kernels():
    """Negatives that draw on more than 2K distinct entities have no shared pool: the scan says so and the general
    kernels score them (same result as with the scan switched off)."""
    from mkb_amd import datasets, models
    from mkb_amd.sampling.negative_sampling import PoolInfo
    import mkb_amd.models.base as model_base

    ds = datasets.Fb15k237(batch_size=512, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(1)
    m = models.TransE(hidden_dim=64, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
    s = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)[:512]).cuda()
    neg = torch.randint(0, len(ds.entities), (512, 64), generator=torch.Generator().manual_seed(3)).cuda()
    assert PoolInfo.discover(neg, s, 2) is None
    with torch.no_grad():
        a = m(s, neg, "tail-batch")
        model_base.AUTO_POOL = False
        try:
            b = m(s, neg, "tail-batch")
        finally:
            model_base.AUTO_POOL = True
    assert torch.equal(a, b)

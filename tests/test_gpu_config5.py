"""-m gpu: BASELINE config 5's workload -- datasets.Yago310 (123,182 entities, 37 relations; the training triples are the
documented SYNTHETIC stand-in, train.csv is absent upstream) + RotatE hidden 500, K = 256, B = 1024.
  (a) the on-device sampler is bit-exact with the plain-C oracle on this graph (Zipf-heavy true sets: the bitmap / Bloom /
      sort branches of the filter all fire);
  (b) the fused step equals the oracle at reduced dim (the 493 MB table makes a full-dim oracle step minutes long);
  (c) at full dim the pooled kernels equal the general kernels (independent implementation) on scores, loss and both
      dense gradients, and a 48-row slice of the same batch equals the oracle (zero weight elsewhere: see
      test_gpu_pool.py::test_full_size_fused_step_gradients_vs_oracle_on_a_128_row_slice)."""
import ctypes
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util_gpu import grad_close  # noqa: E402  (tests/ is on sys.path: conftest.py)
ATOL = 1e-4


@pytest.fixture(scope="module")
def yago():
    from mkb_amd import datasets

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        ds = datasets.Yago310(batch_size=1024, shuffle=False, seed=42, num_workers=0)
    assert (ds.n_entity, ds.n_relation) == (123182, 37) and ds.synthetic_train
    return ds, np.asarray(ds.train, dtype=np.int64)


def test_yago310_warns_that_its_training_triples_are_synthetic():
    from mkb_amd import datasets

    with pytest.warns(RuntimeWarning, match="SYNTHETIC"):
        datasets.Yago310(batch_size=8, shuffle=False, seed=42, num_workers=0)


def test_sampler_bit_exact_vs_c_oracle(yago, liboracle):
    from mkb_amd import sampling
    from mkb_amd.sampling.negative_sampling import _filter_csr

    ds, train_np = yago
    train = torch.as_tensor(train_np).cuda()
    K, B, N, R = 256, 1024, ds.n_entity, ds.n_relation
    ns = sampling.NegativeSampling(size=K, train_triples=train_np, entities=ds.entities, relations=ds.relations, seed=11)
    (hk, ho, hv, _), (tk, to, tv, _) = _filter_csr(train_np, N, R)
    st = ctypes.create_string_buffer(4 * 624 + 4)
    liboracle.orc_mt_seed(st, ctypes.c_uint32(11))
    liboracle.orc_generate.restype = ctypes.c_int
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    pick = np.random.RandomState(3)
    for c in range(8):
        idx = pick.randint(len(train_np), size=B)
        head = c % 2 == 0
        got = ns.generate(train[torch.as_tensor(idx).cuda()], "head-batch" if head else "tail-batch")
        smp = np.ascontiguousarray(train_np[idx])
        want, pool = np.zeros((B, K), dtype=np.int64), np.zeros(2 * K, dtype=np.int64)
        k, o, v, stride = (hk, ho, hv, N) if head else (tk, to, tv, R)
        rc = liboracle.orc_generate(st, ctypes.c_int64(N), ctypes.c_int64(K), p(smp), ctypes.c_int64(B), ctypes.c_int(head),
                                    p(k), ctypes.c_int64(len(k)), p(o), p(v), ctypes.c_int64(stride), p(want), p(pool))
        assert rc == 0
        np.testing.assert_array_equal(got._mkb_pool.pool.cpu().numpy(), pool)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    ns.check()


def test_fused_step_vs_oracle_reduced_dim(yago):
    from mkb_amd import models, sampling
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring

    ds, train_np = yago
    B, K, hidden = 96, 256, 12
    torch.manual_seed(3)
    m = models.RotatE(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=6.0)
    tb = scoring.Tables("RotatE", hidden, 6.0, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone())
    m = m.cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=train_np, entities=ds.entities, relations=ds.relations, seed=42)
    step = FusedTrainStep(m, alpha=0.5)
    pick = np.random.RandomState(7)
    for mode in ("head-batch", "tail-batch"):
        s = torch.as_tensor(train_np[pick.randint(len(train_np), size=B)]).cuda()
        w = (torch.rand(B) + 0.1).cuda()
        m.zero_grad(set_to_none=True)
        neg = ns.generate(s, mode)
        loss = step(s, w, neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu(), neg.cpu(), w.cpu(), mode, 0.5, fast_norm=True)
        np.testing.assert_allclose(step.negative_score.cpu().numpy(), ref["neg"].numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
    ns.check()


def test_full_dim_pooled_vs_general_and_oracle_slice(yago, monkeypatch):
    import mkb_amd.models.base as model_base
    from mkb_amd import losses, models, sampling
    from mkb_amd.fused import FusedTrainStep
    from oracle import scoring

    monkeypatch.setattr(model_base, "AUTO_POOL", False)
    ds, train_np = yago
    B, K, hidden = 1024, 256, 500
    torch.manual_seed(4)
    m = models.RotatE(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=6.0)
    tb = scoring.Tables("RotatE", hidden, 6.0, m.entity_embedding.detach().clone(), m.relation_embedding.detach().clone(),
                        m.modulus.detach().clone())
    m = m.cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=train_np, entities=ds.entities, relations=ds.relations, seed=42)
    s = torch.as_tensor(train_np[np.random.RandomState(5).randint(len(train_np), size=B)]).cuda()
    w = (torch.rand(B) + 0.1).cuda()
    rows = torch.cat([torch.arange(24), torch.arange(24, B, 41)[:24]])
    w_slice = torch.zeros(B)
    w_slice[rows] = torch.rand(len(rows)) + 0.1
    for mode in ("head-batch", "tail-batch"):
        neg = ns.generate(s, mode)
        step = FusedTrainStep(m, alpha=0.5)
        m.zero_grad(set_to_none=True)
        loss = step(s, w, neg, mode)
        neg_f = step.negative_score.clone()
        g_f = (m.entity_embedding.grad.clone(), m.relation_embedding.grad.clone())
        m.zero_grad(set_to_none=True)
        pos_g, neg_g = m(s), m(s, neg.clone(), mode)  # a plain copy has no pool description: general kernels
        err = losses.Adversarial(alpha=0.5)(pos_g, neg_g, w)
        err.backward()
        np.testing.assert_allclose(neg_f.cpu().numpy(), neg_g.detach().cpu().numpy(), rtol=0, atol=ATOL)
        np.testing.assert_allclose(loss.item(), err.item(), rtol=0, atol=1e-5)
        grad_close(g_f[0].cpu().numpy(), m.entity_embedding.grad.cpu().numpy(), rel=1e-5)
        grad_close(g_f[1].cpu().numpy(), m.relation_embedding.grad.cpu().numpy(), rtol=1e-4, rel=1e-5)
        # oracle on the slice that carries the weight
        m.zero_grad(set_to_none=True)
        loss = step(s, w_slice.cuda(), neg, mode)
        ref = scoring.train_step_grads(tb, s.cpu()[rows], neg.cpu()[rows], w_slice[rows], mode, 0.5, fast_norm=True)
        np.testing.assert_allclose(loss.item(), ref["loss"].item(), rtol=0, atol=1e-5)
        grad_close(m.entity_embedding.grad.cpu().numpy(), ref["g_ent"].numpy())
        grad_close(m.relation_embedding.grad.cpu().numpy(), ref["g_rel"].numpy(), rtol=1e-4)
    ns.check()

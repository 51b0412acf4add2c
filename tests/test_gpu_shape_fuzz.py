"""-m gpu: random shapes through both routes.  For each seed: a model family, hidden size, batch size and negative count are
drawn (odd sizes, sizes just above / below the kernels' tile edges included), the step's scores, loss and BOTH dense gradients
are computed by the pooled route (shared-pool kernels: tile / single-pass / matrix-core forms, whichever the shape selects), by the fused
step (one library call with the row kernels and the loss rows folded in) and by the general route (arbitrary-candidate kernels + autograd), and compared on every element with tolerances relative to the
reference's scale.  The row clamp of the 128-row GEMM tile (rounds 3-4) was a tile-edge bug of exactly the kind a fixed list of
shapes does not meet."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import test_gpu_pool as T  # noqa: E402  (tests/ is on sys.path: conftest.py)

EDGES = [1, 2, 3, 7, 8, 9, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025]


def _draw(seed):
    rs = np.random.RandomState(1000 + seed)
    name = T.MODELS[seed % len(T.MODELS)]

    def pick(lo, hi):
        if rs.rand() < 0.5:
            c = [e for e in EDGES if lo <= e <= hi]
            return int(c[rs.randint(len(c))])
        return int(rs.randint(lo, hi + 1))

    hidden = pick(2, 1100)
    B = pick(1, 1300)
    K = pick(1, 256)
    return name, hidden, B, K


@pytest.mark.parametrize("seed", range(int(os.environ.get("MKB_FUZZ_SEEDS", "40"))))  # (MKB_FUZZ_SEEDS=400: the long hunt)
def test_random_shape_pooled_equals_general(seed):
    import mkb_amd.models.base as mb
    from mkb_amd import losses
    from mkb_amd.fused import pooled_supported

    name, hidden, B, K = _draw(seed)
    ds, m, tb, ns, train = T._setup("Fb15k237", name, hidden, B, K, gamma=9.0, seed=seed)
    if not pooled_supported(m, B, K):
        pytest.skip(f"{name} hidden {hidden} B {B} K {K}: not a pooled shape")
    idx = torch.as_tensor(np.random.RandomState(seed).randint(len(train), size=B))
    s = train[idx].cuda()
    w = (torch.rand(B, generator=torch.Generator().manual_seed(seed)) + 0.1).cuda()
    mode = "head-batch" if seed % 2 == 0 else "tail-batch"
    neg = ns.generate(s, mode)
    plain = neg.clone()
    got = {}
    for tag, n in (("pooled", neg), ("general", plain)):
        m.zero_grad(set_to_none=True)
        mb.AUTO_POOL = tag == "pooled"
        try:
            sc = m(s, n, mode)
        finally:
            mb.AUTO_POOL = True
        err = losses.Adversarial(alpha=1.0)(m(s), sc, w)
        err.backward()
        got[tag] = (sc.detach().cpu().numpy(), err.item(), m.entity_embedding.grad.cpu().numpy().copy(),
                    m.relation_embedding.grad.cpu().numpy().copy())
    # ... and the fused step (one library call: row kernels + pooled forward + loss rows + single-pass / matrix backward)
    from mkb_amd.fused import FusedTrainStep

    m.zero_grad(set_to_none=True)
    step = FusedTrainStep(m, alpha=1.0)
    loss = step(s, w, neg, mode)
    got["fused"] = (step.negative_score.cpu().numpy(), loss.item(), m.entity_embedding.grad.cpu().numpy().copy(),
                    m.relation_embedding.grad.cpu().numpy().copy())
    what = f"{name} hidden {hidden} B {B} K {K} {mode}"
    sscale = max(np.abs(got["general"][0]).max(), 1e-6)
    for route in ("pooled", "fused"):
        tag = f"{what} [{route}]"
        np.testing.assert_allclose(got[route][0], got["general"][0], rtol=0, atol=2e-5 * max(sscale, 1.0), err_msg=tag)
        assert abs(got[route][1] - got["general"][1]) <= 2e-6 * max(1.0, abs(got["general"][1])), tag
        for k in (2, 3):
            scale = np.abs(got["general"][k]).max()
            np.testing.assert_allclose(got[route][k], got["general"][k], rtol=0, atol=2e-5 * max(scale, 1e-30), err_msg=tag)
    ns.check()


_N_RANK = max(1, int(os.environ["MKB_FUZZ_SEEDS"]) // 4) if "MKB_FUZZ_SEEDS" in os.environ else 10  # (each case walks 14,541 candidates per query on the host)


@pytest.mark.parametrize("seed", range(_N_RANK))
def test_random_shape_device_ranking_equals_reference_route(seed):
    """The filtered ranking (mkb_rank: tiled all-entity forward + on-device filter count) against the reference's route
    (TestDataset rows + general forward + argsort) for a random model / hidden size / evaluation batch size on FB15k-237's
    14,541 candidates.  Single ranks may swap where two candidates' scores are closer than the routes' summation noise, so the
    means are compared closely rather than exactly (a mis-addressed tile moves hundreds of ranks per query)."""
    from mkb_amd import datasets, evaluation, models

    rs = np.random.RandomState(4000 + seed)
    name = T.MODELS[seed % len(T.MODELS)]
    hidden = int(rs.choice([e for e in EDGES if 2 <= e <= 600])) if rs.rand() < 0.5 else int(rs.randint(2, 601))
    bs = int(rs.randint(1, 71))
    ds = datasets.Fb15k237(batch_size=8, shuffle=False, seed=42, num_workers=0)
    torch.manual_seed(seed)
    m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=9).cuda().eval()
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=bs,
                               device="cuda", num_workers=0)
    lo = int(rs.randint(0, len(ds.test) - 40))
    test = ds.test[lo: lo + 1 + int(rs.randint(1, 40))]
    fast = ev.eval(model=m, dataset=test)
    ev.force_reference_path = True
    slow = ev.eval(model=m, dataset=test)
    what = f"{name} hidden {hidden} eval batch {bs} triples {len(test)}: {fast} vs {slow}"
    assert abs(fast["MR"] - slow["MR"]) <= 0.002 * slow["MR"] + 0.5, what
    assert abs(fast["MRR"] - slow["MRR"]) <= 2e-3, what
    for k in ("HITS@1", "HITS@3", "HITS@10"):
        assert abs(fast[k] - slow[k]) <= 1.01 / (2 * len(test)) + 1e-9, what

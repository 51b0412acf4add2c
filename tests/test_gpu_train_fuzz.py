"""-m gpu: random shapes through the whole training loop.  For each seed a model family, hidden size, batch size and negative count
are drawn; eight steps are trained twice from the same tables:
  (a) the product's fast loop -- FusedTrainStep.sampled (sampler riding the optimizer's catch-up launch, pooled kernels, loss rows)
      with mkb_amd.optim.Adam(lazy_rows=True, draw_ahead=sampler, defer_step=True) -- and a closing flush;
  (b) the reference's loop as a user would write it (README.md:448-474): sampler.generate, model(sample), model(sample, negatives,
      mode) on the general kernels, losses.Adversarial, backward, torch.optim.Adam.step on DENSE gradients.
Both must end in the same tables: exact dense Adam semantics of the row-lazy optimizer (every row decays every step), identical
negatives, identical scores / loss / gradients up to summation order.  FB15k-237's 14,541 entities (row-lazy path), real triples."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import test_gpu_pool as T  # noqa: E402  (tests/ is on sys.path: conftest.py)
from test_gpu_shape_fuzz import EDGES  # noqa: E402


def _draw(seed):
    rs = np.random.RandomState(7000 + seed)
    name = T.MODELS[seed % len(T.MODELS)]

    def pick(lo, hi):
        if rs.rand() < 0.5:
            c = [e for e in EDGES if lo <= e <= hi]
            return int(c[rs.randint(len(c))])
        return int(rs.randint(lo, hi + 1))

    return name, pick(2, 300), pick(1, 700), pick(1, 128)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MKB_FUZZ_SEEDS", "15"))))
def test_random_shape_fast_loop_equals_the_plain_loop(seed):
    import mkb_amd.models.base as mb
    from mkb_amd import datasets, losses, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep, pooled_supported

    name, hidden, B, K = _draw(seed)
    ds = datasets.Fb15k237(batch_size=B, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    w_all = (torch.rand(8 * B, generator=torch.Generator().manual_seed(seed)) + 0.1).cuda()
    lr, steps = 1e-3, 8

    def make():
        torch.manual_seed(seed)
        m = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
        ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
        return m, ns, [p for p in m.parameters() if p.requires_grad]

    m, ns, params = make()
    if not pooled_supported(m, B, K):
        pytest.skip(f"{name} hidden {hidden} B {B} K {K}: not a pooled shape")
    opt = optim.Adam(params, lr=lr, lazy_rows=True, draw_ahead=ns, defer_step=True)
    step = FusedTrainStep(m, alpha=1.0)
    fast_losses = []
    for it in range(steps):
        s = train[it * B: (it + 1) * B].contiguous()
        fast_losses.append(step.sampled(s, w_all[it * B: (it + 1) * B].contiguous(), ns, "head-batch" if it % 2 == 0 else "tail-batch"))
        opt.step()
        opt.zero_grad()
    opt.flush()
    ns.check()
    fast = [p.detach().clone() for p in params] + [torch.stack(fast_losses)]

    m, ns, params = make()
    ref = torch.optim.Adam(params, lr=lr)
    loss_fn = losses.Adversarial(alpha=1.0)
    plain_losses = []
    mb.AUTO_POOL = False  # the general kernels, whatever the negatives look like
    try:
        for it in range(steps):
            s = train[it * B: (it + 1) * B].contiguous()
            mode = "head-batch" if it % 2 == 0 else "tail-batch"
            neg = ns.generate(s, mode).clone()
            err = loss_fn(m(s), m(s, neg, mode), w_all[it * B: (it + 1) * B].contiguous())
            ref.zero_grad()
            err.backward()
            ref.step()
            plain_losses.append(err.detach())
    finally:
        mb.AUTO_POOL = True
    plain = [p.detach().clone() for p in params] + [torch.stack(plain_losses)]

    what = f"{name} hidden {hidden} B {B} K {K}"
    np.testing.assert_allclose(fast[-1].cpu().numpy(), plain[-1].cpu().numpy(), rtol=2e-6, atol=5e-6, err_msg=what)  # (losses up to ~100: a few fp32 ulp)
    # an element moves <= lr per step.  Gradients are sums in different orders (and fp32 atomics): where a sum cancels to exactly
    # 0 in one order and to a rounding residue in the other, Adam turns the residue into a full step (tests/test_gpu_defer.py) --
    # so: nearly every element agrees to a small fraction of one step, none is off by more than the steps taken
    for a, b in zip(fast[:-1], plain[:-1]):
        diff = (a - b).abs()
        n_off = int((diff > 0.02 * lr).sum())
        assert n_off <= max(64, a.numel() // 2000) and float(diff.max()) <= steps * lr * 1.01, (what, n_off, float(diff.max()))

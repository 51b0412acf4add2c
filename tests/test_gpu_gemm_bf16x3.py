"""The bf16 matrix-pipe form of the pooled GEMMs (gemm128_bf16x3_mfma_kernel, gemm_mfma.h): fp32 operands split into three bf16
numbers each, six bf16 products per fp32 product.  It is the default of the GEMM-shaped models (ComplEx, DistMult); the parity
cases of tests/test_gpu_pool.py therefore already run through it against the oracle.  Here: the two forms of the kernel on the
same inputs (MKB_GEMM_BF16X3 is read per call), and the edge shapes of the staging code (K ranges that end inside a chunk, rows
and columns that end inside a tile, depth cuts)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import test_gpu_pool as T  # noqa: E402  (tests/ is on sys.path: conftest.py)


def _grads(name, hidden, B, K, flag, mode, seed=3, no_mfma=False):
    from mkb_amd.fused import FusedTrainStep

    ds, m, tb, ns, train = T._setup("Fb15k237", name, hidden, B, K, gamma=9.0)
    idx = torch.as_tensor(np.random.RandomState(seed).randint(len(train), size=B))
    s = train[idx].cuda()
    w = (torch.rand(B, generator=torch.Generator().manual_seed(seed)) + 0.1).cuda()
    neg = ns.generate(s, mode)
    old = os.environ.get("MKB_GEMM_BF16X3")
    os.environ["MKB_GEMM_BF16X3"] = flag
    if no_mfma:
        os.environ["MKB_POOL_NO_MFMA"] = "1"  # the lane-owns-dims VALU kernels instead of the matrix cores (read per call)
    try:
        m.zero_grad(set_to_none=True)
        loss = FusedTrainStep(m, alpha=1.0)(s, w, neg, mode)
        return float(loss), m.entity_embedding.grad.cpu().numpy().copy(), m.relation_embedding.grad.cpu().numpy().copy()
    finally:
        os.environ.pop("MKB_POOL_NO_MFMA", None)
        if old is None:
            os.environ.pop("MKB_GEMM_BF16X3", None)
        else:
            os.environ["MKB_GEMM_BF16X3"] = old


@pytest.mark.parametrize("name", ["ComplEx", "DistMult"])
@pytest.mark.parametrize("mode", ["head-batch", "tail-batch"])
def test_bf16x3_equals_the_fp32_matrix_instruction_on_the_headline_shape(name, mode):
    """hidden 1000, K 256, B 1024: scores are sums of 2000 / 1000 products of magnitude ~1e-4; the three dropped cross terms of
    the split are below 2^-24 of a product: the two forms agree to ~1e-7 of the gradients' scale."""
    l1, e1, r1 = _grads(name, 1000, 1024, 256, "1", mode)
    l0, e0, r0 = _grads(name, 1000, 1024, 256, "0", mode)
    assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0))  # (fp32 sums of 1024 row terms in two orders: a few ulp)
    scale_e, scale_r = np.abs(e0).max(), np.abs(r0).max()
    assert scale_e > 0 and scale_r > 0
    np.testing.assert_allclose(e1, e0, rtol=0, atol=2e-6 * scale_e)
    np.testing.assert_allclose(r1, r0, rtol=0, atol=2e-6 * scale_r)


@pytest.mark.parametrize("name,hidden,B,K", [
    # (the 128-row tile kernels take products of >= 24 tiles with 4-aligned sizes; smaller ones use the 64 x 64 fp32 kernel)
    ("ComplEx", 1030, 1000, 256),   # De 2060 = 64 chunks + 12: the last chunk of the score product's K range is partial; 2060 columns
                                    # end inside a column tile of dQ / dX; 1000 rows end inside a row tile
    ("DistMult", 1500, 1020, 250),  # 500 pool positions: dQ's K range ends inside a chunk, dX's row tiles end inside a tile
    ("ComplEx", 500, 2048, 384),    # 768 pool positions: the depth cuts drop whole tiles and shorten K ranges
])
def test_bf16x3_edge_shapes_equal_the_fp32_form(name, hidden, B, K):
    for mode in ("head-batch", "tail-batch"):
        l1, e1, r1 = _grads(name, hidden, B, K, "1", mode, seed=5)
        l0, e0, r0 = _grads(name, hidden, B, K, "0", mode, seed=5)
        assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0))  # (fp32 sums of 1024 row terms in two orders: a few ulp)
        np.testing.assert_allclose(e1, e0, rtol=0, atol=2e-6 * max(np.abs(e0).max(), 1e-30))
        np.testing.assert_allclose(r1, r0, rtol=0, atol=2e-6 * max(np.abs(r0).max(), 1e-30))


@pytest.mark.parametrize("name", ["ComplEx", "DistMult"])
@pytest.mark.parametrize("flag", ["1", "0"])
def test_matrix_route_equals_the_valu_route_on_every_row(name, flag):
    """Both forms of the 128-row tile kernel against the VALU kernels of the pooled path, EVERY row of the table gradient at
    the headline shape.  (The oracle comparisons of tests/test_gpu_pool.py at this size look at a 128-row slice of the batch;
    rounds 3-4 computed the last three batch rows of the score and dQ products from row B - 4 -- a row clamp meant for the
    row-contiguous operand layout -- and nothing looked there.)"""
    for mode in ("head-batch", "tail-batch"):
        l1, e1, r1 = _grads(name, 1000, 1024, 256, flag, mode)
        l0, e0, r0 = _grads(name, 1000, 1024, 256, flag, mode, no_mfma=True)
        assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0))  # (fp32 sums of 1024 row terms in two orders: a few ulp)
        np.testing.assert_allclose(e1, e0, rtol=0, atol=3e-6 * np.abs(e0).max())
        np.testing.assert_allclose(r1, r0, rtol=0, atol=3e-6 * np.abs(r0).max())


def test_training_with_either_matrix_form_reaches_the_same_model():
    """40 fused steps of ComplEx hidden 1000 (FB15k-237, K 256, B 1024) with the row-lazy optimizer, once with each form of the
    tile kernel: the same loss trajectory and the same tables (Adam divides a gradient by its own running magnitude, so the
    forms' ~1e-7 relative differences stay ~1e-7 of a step)."""
    from mkb_amd import datasets, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep

    def run(flag):
        old = os.environ.get("MKB_GEMM_BF16X3")
        os.environ["MKB_GEMM_BF16X3"] = flag
        try:
            ds = datasets.Fb15k237(batch_size=1024, shuffle=False, seed=42, num_workers=0)
            torch.manual_seed(42)
            m = models.ComplEx(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
            ns = sampling.NegativeSampling(size=256, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
            opt = optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-3, lazy_rows=True)
            step = FusedTrainStep(m, alpha=1.0)
            train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
            w = torch.ones(1024, device="cuda")
            losses = []
            for it in range(40):
                s = train[it * 1024: (it + 1) * 1024].contiguous()
                losses.append(step.sampled(s, w, ns, "head-batch" if it % 2 == 0 else "tail-batch"))
                opt.step()
                opt.zero_grad()
            opt.flush()
            return torch.stack(losses).cpu().numpy(), m.entity_embedding.detach().cpu().numpy(), m.relation_embedding.detach().cpu().numpy()
        finally:
            if old is None:
                os.environ.pop("MKB_GEMM_BF16X3", None)
            else:
                os.environ["MKB_GEMM_BF16X3"] = old

    l1, e1, r1 = run("1")
    l0, e0, r0 = run("0")
    assert l0[-1] < l0[0] - 1e-4, (l0[0], l0[-1])  # (it trains)
    np.testing.assert_allclose(l1, l0, rtol=0, atol=5e-6)
    # lr = 5e-3: an element moves ~5e-3 per step it is touched; fp32 atomics order the sums differently run to run, and Adam
    # turns a sum that cancels to exactly 0 in one order and to a residue in the other into a full step (tests/test_gpu_defer.py)
    for a, b in ((e1, e0), (r1, r0)):
        diff = np.abs(a - b)
        assert int((diff > 1e-4).sum()) <= 64 and float(diff.max()) < 5e-2, (int((diff > 1e-4).sum()), float(diff.max()))


def test_switching_routes_in_one_process_on_dirty_device_memory():
    """The per-call switches change the workspace layout while callers cache one workspace per table shape: the size the
    library reports must cover every setting.  Freed device memory is filled with NaN first, so that a route reading (or a test
    trusting) bytes it never wrote shows up -- in round 4 exactly this sequence (matrix route, then MKB_POOL_NO_MFMA=1 on the
    same shape) ran past a workspace sized for the matrix route: a memory access fault on dirty memory, silence on fresh pages
    (tools/_poison_check.py runs the long form over all five models)."""
    def poison():
        blocks = [torch.full((128 << 20,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(6)]  # 3 GB
        torch.cuda.synchronize()
        del blocks

    ref = _grads("DistMult", 1000, 1024, 256, "1", "head-batch")
    poison()
    a = _grads("DistMult", 1000, 1024, 256, "1", "head-batch")
    poison()
    b = _grads("DistMult", 1000, 1024, 256, "1", "head-batch", no_mfma=True)
    poison()
    c = _grads("DistMult", 1000, 1024, 256, "0", "head-batch")
    for got in (a, b, c):
        assert np.isfinite(got[1]).all() and np.isfinite(got[2]).all()
        assert abs(got[0] - ref[0]) <= 1e-6
        np.testing.assert_allclose(got[1], ref[1], rtol=0, atol=3e-6 * np.abs(ref[1]).max())
        np.testing.assert_allclose(got[2], ref[2], rtol=0, atol=3e-6 * np.abs(ref[2]).max())


def test_workspace_guard_notices_a_write_behind_the_workspace(monkeypatch):
    """MKB_WS_GUARD=1 puts 64 KB of a byte pattern behind every cached pooled-kernel workspace (the library takes the workspace
    as a bare pointer); the conftest fixture checks it after every GPU test.  Here: the check itself."""
    from mkb_amd import datasets, fused, models

    monkeypatch.setenv("MKB_WS_GUARD", "1")
    ds = datasets.Umls(batch_size=8, shuffle=False, seed=42, num_workers=0)
    m = models.DistMult(hidden_dim=44, entities=ds.entities, relations=ds.relations, gamma=6).cuda()
    ws = fused._workspace(m, 96, 24)  # (a shape no other test uses: a fresh cache entry, with its guard)
    fused.check_workspace_guards()
    key = [k for k in fused._guards if k[6:8] == (96, 24) and k[3] == m.entity_dim][0]  # (device, stream, name, entity_dim, n_entity, n_relation, B, K, slot)
    guard = fused._guards[key]
    assert guard.data_ptr() == ws.data_ptr() + ws.numel()
    guard[5] = 0
    with pytest.raises(RuntimeError, match="workspace overrun"):
        fused.check_workspace_guards()
    guard[5] = fused._GUARD_VALUE
    fused.check_workspace_guards()


@pytest.mark.parametrize("name,hidden,B,K", [("ComplEx", 1000, 1024, 256), ("DistMult", 1000, 1024, 256), ("ComplEx", 1030, 1000, 256),
                                             ("DistMult", 1500, 1020, 250), ("ComplEx", 500, 2048, 384)])
def test_backward_products_in_one_launch_equal_two_launches(name, hidden, B, K, monkeypatch):
    """Round 5: dQ = G . X and dX = G^T . Q of the bilinear models ride ONE launch with their workgroups interleaved
    (gemm128_bf16x3_pair_kernel).  The same tile code computes the same tiles in the same order per element, so the gradients
    must equal those of the two-launch form (MKB_GEMM_NO_PAIR=1, read per call) up to the order of the final atomics -- at the
    headline shape and at the edge shapes of the staging code (partial chunks, partial tiles, depth cuts)."""
    for mode in ("head-batch", "tail-batch"):
        monkeypatch.delenv("MKB_GEMM_NO_PAIR", raising=False)
        a = _grads(name, hidden, B, K, "1", mode)
        monkeypatch.setenv("MKB_GEMM_NO_PAIR", "1")
        b = _grads(name, hidden, B, K, "1", mode)
        monkeypatch.delenv("MKB_GEMM_NO_PAIR", raising=False)
        assert abs(a[0] - b[0]) <= 1e-6 * max(1.0, abs(b[0]))
        np.testing.assert_allclose(a[1], b[1], rtol=0, atol=1e-6 * np.abs(b[1]).max())
        np.testing.assert_allclose(a[2], b[2], rtol=0, atol=1e-6 * np.abs(b[2]).max())

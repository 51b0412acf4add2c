"""The DENSE pass of the single-pass pooled backward (pool_bwd1_kernel<..., DENSE>): the parity cases of
tests/test_gpu_pool.py that reach the single-pass backward, re-run with MKB_POOL_DENSE=1.  The dense form is compiled for the
complex-modulus pair function (RotatE) only -- there it is the default, the flag changes nothing; the real-valued models have
no dense form since round 4 and the TransE / pRotatE cases below check that asking for it is harmless (general pass, same
gradients).  The switch is read by the library at every call, so it can be flipped inside one process."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import test_gpu_pool as T  # noqa: E402  (tests/ is on sys.path: conftest.py)


@pytest.fixture(autouse=True)
def _dense_on(monkeypatch):
    monkeypatch.setenv("MKB_POOL_DENSE", "1")
    yield


@pytest.mark.parametrize("name", ["RotatE", "TransE", "pRotatE"])
@pytest.mark.parametrize("mode", ["head-batch", "tail-batch"])
def test_dense_pass_pooled_forward_backward_equals_general_and_oracle(name, mode):
    """Umls, B = 77 (a ragged last row tile, waves without a tile: the hand-off ring must stay closed), K = 16."""
    T.test_pooled_forward_backward_equals_general_and_oracle(name, mode)


@pytest.mark.parametrize("cls,name,hidden,B,K", [
    ("Umls", "TransE", 64, 256, 16),
    ("Wn18rr", "RotatE", 32, 200, 128),
    ("Fb15k237", "RotatE", 40, 160, 256),
    ("Fb15k237", "pRotatE", 24, 96, 256),
    ("Fb15k237", "RotatE", 260, 100, 96),
    ("Wn18rr", "RotatE", 512, 67, 64),
])
def test_dense_pass_fused_step_vs_oracle_real_graphs(cls, name, hidden, B, K):
    T.test_fused_step_vs_oracle_real_graphs(cls, name, hidden, B, K)


@pytest.mark.parametrize("name", ["RotatE", "TransE"])
def test_dense_pass_full_size_fused_step_gradients_vs_oracle_on_a_128_row_slice(name):
    """Headline shape (FB15k-237, hidden 1000, K 256, B 1024): dense lanes 32 of 64 per half, four dense positions per phase."""
    T.test_full_size_fused_step_gradients_vs_oracle_on_a_128_row_slice(name)


def test_dense_pass_config2_full_size():
    T.test_config2_full_size_fused_step_vs_oracle()
    T.test_config2_full_size_pooled_equals_general()


@pytest.mark.parametrize("name,hidden,B,K", [
    ("RotatE", 250, 2048, 384),    # 768 positions / 4 blocks = 192 per block = 3 halves of 64 (rounded up to 4), dense lanes 16
    ("TransE", 500, 2048, 384),
    ("TransE", 201, 4096, 300),
])
def test_dense_pass_position_blocks_not_a_power_of_two(name, hidden, B, K):
    T.test_single_pass_backward_position_blocks_not_a_power_of_two(name, hidden, B, K)


@pytest.mark.parametrize("name", ["RotatE", "TransE"])
@pytest.mark.parametrize("B,K,hidden", [(1, 3, 6), (13, 5, 33), (9, 7, 130)])
def test_dense_pass_edge_shapes_with_wrapping_rows(name, B, K, hidden):
    T.test_fused_step_edge_shapes_with_wrapping_rows(name, B, K, hidden)


def test_dense_pass_equals_general_pass_on_the_headline_shape():
    """The two forms of the kernel, same inputs: the dense positions are summed in the same chunk rotation as before, the
    fringe in row order by its slot's owner -- float addition order differs, so the comparison is to 1e-6, not bitwise."""
    import os

    from mkb_amd.fused import FusedTrainStep

    ds, m, tb, ns, train = T._setup("Fb15k237", "RotatE", 1000, 1024, 256, gamma=9.0)
    idx = torch.as_tensor(np.random.RandomState(3).randint(len(train), size=1024))
    s = train[idx].cuda()
    w = (torch.rand(1024) + 0.1).cuda()
    neg = ns.generate(s, "tail-batch")
    grads = []
    for flag in ("1", "0"):
        os.environ["MKB_POOL_DENSE"] = flag
        m.zero_grad(set_to_none=True)
        FusedTrainStep(m, alpha=1.0)(s, w, neg, "tail-batch")
        grads.append((m.entity_embedding.grad.cpu().numpy().copy(), m.relation_embedding.grad.cpu().numpy().copy()))
    np.testing.assert_allclose(grads[0][0], grads[1][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(grads[0][1], grads[1][1], rtol=0, atol=1e-6)
    assert np.abs(grads[1][0]).max() > 0

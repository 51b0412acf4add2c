"""Helpers shared by the -m gpu parity tests: build an mkb_amd model on the device from raw tables."""
import numpy as np
import torch

import mkb_amd
from mkb_amd import models
from oracle import scoring

DEV = "cuda"


def make_model(name, ent, rel, hidden, gamma, modulus=None):
    N, R = ent.shape[0], rel.shape[0]
    m = getattr(models, name)(hidden_dim=hidden, entities={i: i for i in range(N)}, relations={i: i for i in range(R)},
                              gamma=gamma)
    with torch.no_grad():
        m.entity_embedding.copy_(torch.as_tensor(ent))
        m.relation_embedding.copy_(torch.as_tensor(rel))
        if modulus is not None and hasattr(m, "modulus"):
            m.modulus.copy_(torch.as_tensor(modulus))
    return m.to(DEV)


def oracle_tables(name, ent, rel, hidden, gamma, modulus=None):
    mod = None
    if name in ("RotatE", "pRotatE"):
        mod = torch.as_tensor(modulus).clone().float() if modulus is not None else None
    return scoring.Tables(name, hidden, gamma, torch.as_tensor(ent).clone().float(), torch.as_tensor(rel).clone().float(), mod)


def random_problem(name, N, R, hidden, B, K, seed, gamma=6.0):
    g = torch.Generator().manual_seed(seed)
    de, dr = scoring.dims(name, hidden)
    rng = (gamma + 2.0) / hidden
    ent = (torch.rand(N, de, generator=g) * 2 - 1) * rng
    rel = (torch.rand(R, dr, generator=g) * 2 - 1) * rng
    sample = torch.stack([torch.randint(N, (B,), generator=g), torch.randint(R, (B,), generator=g),
                          torch.randint(N, (B,), generator=g)], 1)
    neg = torch.randint(N, (B, K), generator=g)
    w = torch.rand(B, generator=g) + 0.1
    modulus = torch.tensor([[0.5 * rng]]) if name in ("RotatE", "pRotatE") else None
    return ent, rel, sample, neg, w, modulus


from util_gpu_tables import headline_tables  # noqa: E402,F401  (re-exported for the GPU tests)


def grad_close(got, ref, rtol=0.0, rel=2e-4):
    """Gradients against a reference: absolute 1e-5, or `rel` of the reference's largest entry where that is tighter (at the
    full-size shapes the entries are ~1e-8 .. 1e-6: an absolute 1e-5 would accept anything)."""
    import numpy as np

    got, ref = np.asarray(got), np.asarray(ref)
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=min(1e-5, rel * max(float(np.abs(ref).max()), 1e-30)))


def dirty_device_memory(scale=1.0):
    """MKB_TEST_DIRTY_MEMORY=<GB>: fill that much (x scale) device memory with NaN and free it (see tests/conftest.py); the
    worker processes of the multi-process tests call this once at their start."""
    import os

    try:
        gb = float(os.environ.get("MKB_TEST_DIRTY_MEMORY", "0") or 0)
    except ValueError:
        gb = 0.0
    if gb > 0 and torch.cuda.is_available():
        n = max(1, int(gb * scale))
        blocks = [torch.full((256 << 20,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        del blocks

"""torch-CPU fp32 restatement of the reference scoring + loss path (TEST INFRASTRUCTURE ONLY).

Same operation order as the reference so that it is the bit-level comparator that can travel
to the GPU box (the reference's Python cannot).  Every function cites the file:line it follows
under ``/root/reference``.

Layout conventions (reference ``mkb/models/base.py:66-100``):
  ent [N, De] fp32 row-major, rel [R, Dr] fp32 row-major, sample [B, 3] int64 (h, r, t),
  negative_sample [B, K] int64, mode in {None, "head-batch", "tail-batch"}.
  RotatE/ComplEx rows: first half = real, second half = imaginary (``torch.chunk``,
  ``rotate.py:76-77``, ``complex.py:70-72``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

MODELS = ("TransE", "RotatE", "ComplEx", "DistMult", "pRotatE")


def dims(model: str, hidden_dim: int):
    """(entity_dim, relation_dim): transe.py:55-63, rotate.py:60-63, complex.py:55-63,
    distmult.py:53-61, protate.py:60-68."""
    if model == "RotatE":
        return 2 * hidden_dim, hidden_dim
    if model == "ComplEx":
        return 2 * hidden_dim, 2 * hidden_dim
    return hidden_dim, hidden_dim


@dataclass
class Tables:
    """Parameter set of one model (base.py:66-100; rotate.py:66-67; protate.py:72)."""

    model: str
    hidden_dim: int
    gamma: float
    ent: torch.Tensor  # [N, De]
    rel: torch.Tensor  # [R, Dr]
    modulus: torch.Tensor | None = None  # [1, 1] for RotatE (unused) and pRotatE (trainable)

    @property
    def embedding_range(self) -> float:
        # base.py:81-84: python double (gamma.item()+2)/hidden_dim stored as fp32, read back by .item()
        g = torch.tensor([self.gamma], dtype=torch.float32).item()
        return torch.tensor([(g + 2.0) / self.hidden_dim], dtype=torch.float32).item()


def init_tables(model: str, n_entity: int, n_relation: int, hidden_dim: int, gamma: float) -> Tables:
    """Same RNG consumption order as base.py:86-100 (entity table first, then relation table),
    then modulus (no RNG) for RotatE / pRotatE."""
    de, dr = dims(model, hidden_dim)
    t = Tables(model, hidden_dim, gamma, torch.zeros(n_entity, de), torch.zeros(n_relation, dr))
    rng = t.embedding_range
    torch.nn.init.uniform_(t.ent, -rng, rng)
    torch.nn.init.uniform_(t.rel, -rng, rng)
    if model in ("RotatE", "pRotatE"):
        t.modulus = torch.tensor([[0.5 * rng]], dtype=torch.float32)
    return t


def format_sample(sample, negative_sample=None):
    """base.py:131-151."""
    if sample.dim() == 2:
        if negative_sample is None:
            return sample, (sample.size(0), 1)
        return sample, tuple(negative_sample.shape)
    return sample.reshape(-1, 3), (sample.size(0), sample.size(1))


def gather(ent, rel, sample, negative_sample, mode):
    """base.py:153-207: materialised [B,1|K,D] operands."""
    if mode == "head-batch":
        b, k = negative_sample.shape
        h = ent.index_select(0, negative_sample.reshape(-1)).view(b, k, -1)
        r = rel.index_select(0, sample[:, 1]).unsqueeze(1)
        t = ent.index_select(0, sample[:, 2]).unsqueeze(1)
    elif mode == "tail-batch":
        b, k = negative_sample.shape
        h = ent.index_select(0, sample[:, 0]).unsqueeze(1)
        r = rel.index_select(0, sample[:, 1]).unsqueeze(1)
        t = ent.index_select(0, negative_sample.reshape(-1)).view(b, k, -1)
    else:
        h = ent.index_select(0, sample[:, 0]).unsqueeze(1)
        r = rel.index_select(0, sample[:, 1]).unsqueeze(1)
        t = ent.index_select(0, sample[:, 2]).unsqueeze(1)
    return h, r, t


def score(tb: Tables, sample, negative_sample=None, mode=None, ent=None, rel=None, modulus=None,
          fast_norm=False):
    """model.forward for the five KGE models.

    ``ent`` / ``rel`` / ``modulus`` override the tables (used to pass autograd leaves).
    ``fast_norm=True`` replaces RotatE's ``stack -> norm(dim=0)`` (rotate.py:95-96, a torch-CPU
    pathology) by ``sqrt(re^2+im^2)``; results agree to 1 ulp-ish and it is
    only used for the second CPU-baseline figure.
    """
    ent = tb.ent if ent is None else ent
    rel = tb.rel if rel is None else rel
    modulus = tb.modulus if modulus is None else modulus
    sample, shape = format_sample(sample, negative_sample)
    h, r, t = gather(ent, rel, sample, negative_sample, mode)
    head_mode = mode == "head-batch"
    m = tb.model
    if m == "TransE":  # transe.py:65-76
        x = h + (r - t) if head_mode else (h + r) - t
        s = _gamma(tb) - torch.norm(x, p=1, dim=2)
    elif m == "DistMult":  # distmult.py:63-75
        x = h * (r * t) if head_mode else (h * r) * t
        s = x.sum(dim=2)
    elif m == "ComplEx":  # complex.py:65-85
        re_h, im_h = torch.chunk(h, 2, dim=2)
        re_r, im_r = torch.chunk(r, 2, dim=2)
        re_t, im_t = torch.chunk(t, 2, dim=2)
        if head_mode:
            re_s = re_r * re_t + im_r * im_t
            im_s = re_r * im_t - im_r * re_t
            x = re_h * re_s + im_h * im_s
        else:
            re_s = re_h * re_r - im_h * im_r
            im_s = re_h * im_r + im_h * re_r
            x = re_s * re_t + im_s * im_t
        s = x.sum(dim=2)
    elif m == "RotatE":  # rotate.py:69-99
        re_h, im_h = torch.chunk(h, 2, dim=2)
        re_t, im_t = torch.chunk(t, 2, dim=2)
        phase = r / (tb.embedding_range / math.pi)
        re_r, im_r = torch.cos(phase), torch.sin(phase)
        if head_mode:
            re_s = re_r * re_t + im_r * im_t
            im_s = re_r * im_t - im_r * re_t
            re_s = re_s - re_h
            im_s = im_s - im_h
        else:
            re_s = re_h * re_r - im_h * im_r
            im_s = re_h * im_r + im_h * re_r
            re_s = re_s - re_t
            im_s = im_s - im_t
        if fast_norm:
            n = torch.sqrt(re_s * re_s + im_s * im_s)
        else:
            n = torch.stack([re_s, im_s], dim=0).norm(dim=0)
        s = _gamma(tb) - n.sum(dim=2)
    elif m == "pRotatE":  # protate.py:74-93
        k = tb.embedding_range / math.pi
        ph, pr, pt = h / k, r / k, t / k
        x = ph + (pr - pt) if head_mode else (ph + pr) - pt
        x = torch.abs(torch.sin(x))
        s = _gamma(tb) - x.sum(dim=2) * modulus
    else:
        raise ValueError(m)
    return s.view(shape)


def _gamma(tb: Tables) -> float:
    # gamma is an fp32 Parameter read back with .item() (base.py:77, transe.py:75)
    return torch.tensor([tb.gamma], dtype=torch.float32).item()


def adversarial(positive_score, negative_score, weight, alpha=0.5):
    """losses/adversarial.py:21-30."""
    ps = F.logsigmoid(positive_score).squeeze(dim=1)
    ns = (F.softmax(negative_score * alpha, dim=1).detach() * F.logsigmoid(-negative_score)).sum(dim=1)
    pl = -(weight * ps).sum() / weight.sum()
    nl = -(weight * ns).sum() / weight.sum()
    return (pl + nl) / 2


def train_step_grads(tb: Tables, sample, negative_sample, weight, mode, alpha, fast_norm=False):
    """One Pipeline inner step up to ``error.backward()`` (compose/pipeline.py:211-236):
    positive forward (mode=None), negative forward (mode), Adversarial, autograd.

    Returns dict(pos, neg, loss, g_ent, g_rel, g_modulus) -- dense gradients like the reference's
    ``index_select`` backward.
    """
    ent = tb.ent.detach().clone().requires_grad_(True)
    rel = tb.rel.detach().clone().requires_grad_(True)
    mod = None
    if tb.modulus is not None:
        mod = tb.modulus.detach().clone().requires_grad_(True)
    pos = score(tb, sample, ent=ent, rel=rel, modulus=mod, fast_norm=fast_norm)
    neg = score(tb, sample, negative_sample, mode, ent=ent, rel=rel, modulus=mod, fast_norm=fast_norm)
    loss = adversarial(pos, neg, weight, alpha)
    loss.backward()
    return {
        "pos": pos.detach(),
        "neg": neg.detach(),
        "loss": loss.detach(),
        "g_ent": ent.grad,
        "g_rel": rel.grad,
        "g_modulus": None if mod is None else mod.grad,  # None for RotatE (unused parameter)
    }


def adam_update(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (single-tensor, no amsgrad/weight decay) as the README loop uses it
    (README.md:123-126).  ``step`` is 1-based.  In-place on p, m, v; dense: every row moves."""
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)
    return p

"""numpy float64 closed-form restatement: scores, Adversarial loss and all gradients
(TEST INFRASTRUCTURE ONLY).

Independent of autograd.  It is written in the *query / candidate* decomposition the HIP
kernels use, so it also documents (and pins) that decomposition:

  every model's score of row i against candidate entity row x is  s = c0 - m * sum_k f(q_i[k], x[k])
  where the per-row *query* q_i is built from the two fixed operands of the triple:

  model     tail-batch / None (candidate = tail)        head-batch (candidate = head)        f(q, x)
  TransE    q = h + r            (transe.py:73)          q = r - t          (transe.py:71)     |q - x| / |x + q|
  DistMult  q = h * r            (distmult.py:71)        q = r * t          (distmult.py:69)   -(q * x)   (c0 = 0)
  ComplEx   q = h (x) r          (complex.py:79-80)      q = conj(r) (x) t  (complex.py:74-75) -(q . x)   (c0 = 0)
  RotatE    q = h (x) e^{i phi}  (rotate.py:89-90)       q = e^{-i phi} (x) t (rotate.py:84-85) |q - x| complex modulus
  pRotatE   q = h/k + r/k        (protate.py:84)         q = r/k - t/k      (protate.py:82)    |sin(q -/+ x/k)|

Gradients follow SURVEY.md section 8 a13 (verified there against reference autograd).
"""
from __future__ import annotations

import math

import numpy as np


def _f64(a):
    return np.asarray(a, dtype=np.float64)


def emb_range_over_pi(gamma: float, hidden_dim: int) -> float:
    """fp32 value of embedding_range (base.py:81-84) divided by pi in double
    (rotate.py:79: ``self.embedding_range.item() / self.pi``)."""
    g = float(np.float32(gamma))
    rng = float(np.float32((g + 2.0) / hidden_dim))
    return rng / math.pi


def build_query(model, ent, rel, sample, head_mode, k):
    """q_i for every row, float64.  ``k`` = embedding_range / pi."""
    ent, rel = _f64(ent), _f64(rel)
    h, r, t = ent[sample[:, 0]], rel[sample[:, 1]], ent[sample[:, 2]]
    if model == "TransE":
        return (r - t) if head_mode else (h + r)
    if model == "DistMult":
        return (r * t) if head_mode else (h * r)
    if model == "ComplEx":
        d = ent.shape[1] // 2
        if head_mode:
            return np.concatenate([r[:, :d] * t[:, :d] + r[:, d:] * t[:, d:],
                                   r[:, :d] * t[:, d:] - r[:, d:] * t[:, :d]], axis=1)
        return np.concatenate([h[:, :d] * r[:, :d] - h[:, d:] * r[:, d:],
                               h[:, :d] * r[:, d:] + h[:, d:] * r[:, :d]], axis=1)
    if model == "RotatE":
        d = ent.shape[1] // 2
        c, s = np.cos(r / k), np.sin(r / k)
        if head_mode:
            return np.concatenate([c * t[:, :d] + s * t[:, d:], c * t[:, d:] - s * t[:, :d]], axis=1)
        return np.concatenate([h[:, :d] * c - h[:, d:] * s, h[:, :d] * s + h[:, d:] * c], axis=1)
    if model == "pRotatE":
        return (r / k - t / k) if head_mode else (h / k + r / k)
    raise ValueError(model)


def pair_scores(model, q, x, head_mode, gamma, k, modulus):
    """s[i, j] for q [B, De] against candidates x [B, K, De] (float64)."""
    q = q[:, None, :]
    if model == "TransE":
        z = (x + q) if head_mode else (q - x)
        return gamma - np.abs(z).sum(-1)
    if model in ("DistMult", "ComplEx"):
        return (q * x).sum(-1)
    if model == "RotatE":
        d = x.shape[-1] // 2
        a, b = q[..., :d] - x[..., :d], q[..., d:] - x[..., d:]
        return gamma - np.sqrt(a * a + b * b).sum(-1)
    if model == "pRotatE":
        z = (x / k + q) if head_mode else (q - x / k)
        return gamma - modulus * np.abs(np.sin(z)).sum(-1)
    raise ValueError(model)


def scores(model, ent, rel, sample, negative_sample, mode, gamma, hidden_dim, modulus=None):
    sample = np.asarray(sample)
    head_mode = mode == "head-batch"
    k = emb_range_over_pi(gamma, hidden_dim)
    g = float(np.float32(gamma))
    q = build_query(model, ent, rel, sample, head_mode, k)
    if negative_sample is None or mode not in ("head-batch", "tail-batch"):
        cand = sample[:, 2:3]  # default_batch: candidate = the true tail (base.py:166-175)
    else:
        cand = np.asarray(negative_sample)
    x = _f64(ent)[cand]
    mod = 0.0 if modulus is None else float(np.asarray(modulus).reshape(-1)[0])
    return pair_scores(model, q, x, head_mode, g, k, mod)


def adversarial(pos, neg, w, alpha):
    """loss and d loss/d pos, d loss/d neg  (adversarial.py:21-30)."""
    pos, neg, w = _f64(pos).reshape(-1), _f64(neg), _f64(w)
    logsig = lambda z: -np.logaddexp(0.0, -z)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    z = alpha * neg
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)
    W = w.sum()
    loss = 0.5 * (-(w * logsig(pos)).sum() / W - (w * (p * logsig(-neg)).sum(1)).sum() / W)
    dpos = -0.5 * (w / W) * sig(-pos)
    dneg = 0.5 * (w / W)[:, None] * p * sig(neg)
    return loss, dpos, dneg


def _accumulate_pair_grads(model, q, x, g, head_mode, k, modulus):
    """Returns (dq [B,De], dx [B,K,De], dmodulus) for upstream g = dL/ds [B,K]."""
    qb = q[:, None, :]
    gb = g[:, :, None]
    dmod = 0.0
    if model == "TransE":
        z = (x + qb) if head_mode else (qb - x)
        sg = np.sign(z)
        dq = -(gb * sg).sum(1)
        dx = -(gb * sg) if head_mode else (gb * sg)
    elif model in ("DistMult", "ComplEx"):
        dq = (gb * x).sum(1)
        dx = gb * qb
    elif model == "RotatE":
        d = x.shape[-1] // 2
        a, b = qb[..., :d] - x[..., :d], qb[..., d:] - x[..., d:]
        n = np.sqrt(a * a + b * b)
        inv = np.where(n > 0, 1.0 / np.where(n > 0, n, 1.0), 0.0)
        u = np.concatenate([a * inv, b * inv], axis=-1)
        dq = -(gb * u).sum(1)
        dx = gb * u
    elif model == "pRotatE":
        z = (x / k + qb) if head_mode else (qb - x / k)
        sz = np.sin(z)
        c = np.cos(z) * np.sign(sz) * modulus
        dq = -(gb * c).sum(1)
        dx = (-(gb * c) if head_mode else (gb * c)) / k
        dmod = -(g * np.abs(sz).sum(-1)).sum()
    else:
        raise ValueError(model)
    return dq, dx, dmod


def _query_backward(model, ent, rel, sample, dq, head_mode, k, g_ent, g_rel):
    """Chain dq through build_query into the dense gradient tables (duplicates add)."""
    ent, rel = _f64(ent), _f64(rel)
    hi, ri, ti = sample[:, 0], sample[:, 1], sample[:, 2]
    h, r, t = ent[hi], rel[ri], ent[ti]
    if model == "TransE":
        if head_mode:
            np.add.at(g_rel, ri, dq); np.add.at(g_ent, ti, -dq)
        else:
            np.add.at(g_ent, hi, dq); np.add.at(g_rel, ri, dq)
    elif model == "DistMult":
        if head_mode:
            np.add.at(g_rel, ri, dq * t); np.add.at(g_ent, ti, dq * r)
        else:
            np.add.at(g_ent, hi, dq * r); np.add.at(g_rel, ri, dq * h)
    elif model == "ComplEx":
        d = ent.shape[1] // 2
        qa, qb = dq[:, :d], dq[:, d:]
        if head_mode:  # q = conj(r) (x) t
            np.add.at(g_rel, ri, np.concatenate([qa * t[:, :d] + qb * t[:, d:], qa * t[:, d:] - qb * t[:, :d]], 1))
            np.add.at(g_ent, ti, np.concatenate([qa * r[:, :d] - qb * r[:, d:], qa * r[:, d:] + qb * r[:, :d]], 1))
        else:  # q = h (x) r
            np.add.at(g_ent, hi, np.concatenate([qa * r[:, :d] + qb * r[:, d:], -qa * r[:, d:] + qb * r[:, :d]], 1))
            np.add.at(g_rel, ri, np.concatenate([qa * h[:, :d] + qb * h[:, d:], -qa * h[:, d:] + qb * h[:, :d]], 1))
    elif model == "RotatE":
        d = ent.shape[1] // 2
        c, s = np.cos(r / k), np.sin(r / k)
        qa, qb = dq[:, :d], dq[:, d:]
        if head_mode:  # q = (c t_re + s t_im, c t_im - s t_re)
            np.add.at(g_ent, ti, np.concatenate([qa * c - qb * s, qa * s + qb * c], 1))
            dphi = qa * (-s * t[:, :d] + c * t[:, d:]) + qb * (-s * t[:, d:] - c * t[:, :d])
        else:  # q = (h_re c - h_im s, h_re s + h_im c)
            np.add.at(g_ent, hi, np.concatenate([qa * c + qb * s, -qa * s + qb * c], 1))
            dphi = qa * (-h[:, :d] * s - h[:, d:] * c) + qb * (h[:, :d] * c - h[:, d:] * s)
        np.add.at(g_rel, ri, dphi / k)
    elif model == "pRotatE":
        if head_mode:
            np.add.at(g_rel, ri, dq / k); np.add.at(g_ent, ti, -dq / k)
        else:
            np.add.at(g_ent, hi, dq / k); np.add.at(g_rel, ri, dq / k)


def train_step_grads(model, ent, rel, sample, negative_sample, weight, mode, alpha, gamma, hidden_dim,
                     modulus=None):
    """Closed-form counterpart of ``oracle.scoring.train_step_grads`` in float64."""
    sample = np.asarray(sample)
    negative_sample = np.asarray(negative_sample)
    k = emb_range_over_pi(gamma, hidden_dim)
    g32 = float(np.float32(gamma))
    mod = 0.0 if modulus is None else float(np.asarray(modulus).reshape(-1)[0])
    entd = _f64(ent)
    head_mode = mode == "head-batch"
    # positive pass: mode None => tail-style formula with the true tail (pipeline.py:211)
    q_pos = build_query(model, ent, rel, sample, False, k)
    x_pos = entd[sample[:, 2:3]]
    pos = pair_scores(model, q_pos, x_pos, False, g32, k, mod)
    q_neg = build_query(model, ent, rel, sample, head_mode, k)
    x_neg = entd[negative_sample]
    neg = pair_scores(model, q_neg, x_neg, head_mode, g32, k, mod)
    loss, dpos, dneg = adversarial(pos, neg, weight, alpha)
    g_ent = np.zeros_like(entd)
    g_rel = np.zeros_like(_f64(rel))
    dq, dx, dm1 = _accumulate_pair_grads(model, q_pos, x_pos, dpos[:, None], False, k, mod)
    np.add.at(g_ent, sample[:, 2], dx[:, 0, :])
    _query_backward(model, ent, rel, sample, dq, False, k, g_ent, g_rel)
    dq, dx, dm2 = _accumulate_pair_grads(model, q_neg, x_neg, dneg, head_mode, k, mod)
    np.add.at(g_ent, negative_sample.reshape(-1), dx.reshape(-1, dx.shape[-1]))
    _query_backward(model, ent, rel, sample, dq, head_mode, k, g_ent, g_rel)
    return {"pos": pos, "neg": neg, "loss": loss, "dpos": dpos, "dneg": dneg,
            "g_ent": g_ent, "g_rel": g_rel, "g_modulus": dm1 + dm2}

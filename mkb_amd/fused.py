"""Pooled (shared-candidate-pool) scoring path and the fused training step.

``pooled_forward``   -- what ``model(sample, negative_sample, mode)`` runs when ``negative_sample`` came from
                        ``mkb_amd.sampling.NegativeSampling`` on the device: ``mkb_pool_score_fwd`` scores every
                        row against the shared pool once, the ``[B, size]`` view is a gather; backward scatters
                        ``d loss / d score`` back onto pool positions and calls ``mkb_pool_score_bwd``.
``FusedTrainStep``   -- one call = lines 211-236 of the reference's compose/pipeline.py (positive forward,
                        negative forward, Adversarial, backward into dense ``.grad``) through ``mkb_pool_step``,
                        bypassing autograd.  Used by ``compose.Pipeline`` when model / sampler / loss are ours.
"""
import torch

from . import _gradshare, _hip, _links
from .sampling.negative_sampling import PoolInfo

__all__ = ["FusedTrainStep", "pooled_forward", "pooled_supported"]

_workspaces = {}


def _workspace(model, B, K, slot=0):
    """Device scratch for the pooled kernels, cached per (device, STREAM, table shape, B, K, slot); ``slot`` separates the
    micro-batches of one step, whose forward scratch must survive until their backward half runs.  Per stream: launches on one
    stream use the scratch one after the other, two models of one shape stepped on two streams (the in-process ranks of
    tests/test_gpu_rows_loopback.py; a user training two models side by side) must not share it -- until round 6 they did."""
    dev = model.entity_embedding.device
    key = (dev, _hip.stream_ptr(dev).value, model.name, model.entity_dim, model.n_entity, model.n_relation, B, K, slot)  # (the size depends on the last six)
    ws = _workspaces.get(key)
    if ws is None:
        n = _hip.lib().mkb_pool_step_workspace_bytes(model._tables(), B, K)
        guard = _GUARD_BYTES if _guard_on() else 0
        buf = torch.empty(n + 256 + guard, dtype=torch.uint8, device=dev)
        off = (-buf.data_ptr()) % 256
        ws = buf[off: off + n]
        if guard:  # (debug: a pattern behind the workspace that no kernel may touch; check_workspace_guards() looks at it)
            buf[off + n:].fill_(_GUARD_VALUE)
            _guards[key] = buf[off + n:]
        _workspaces[key] = ws
    return ws


_GUARD_BYTES, _GUARD_VALUE = 1 << 16, 0xA5
_guards = {}


def _guard_on():
    import os

    return os.environ.get("MKB_WS_GUARD", "0") == "1"


def check_workspace_guards():
    """MKB_WS_GUARD=1 (tests): every cached workspace is followed by 64 KB of a byte pattern; raise if a kernel wrote there
    (the library takes the workspace as a bare pointer: nothing else would notice an overrun that stays inside the allocator's
    block).  Synchronises."""
    for key, g in _guards.items():
        if not bool((g == _GUARD_VALUE).all().item()):
            bad = int((g != _GUARD_VALUE).nonzero()[0].item())
            raise RuntimeError(f"pooled-kernel workspace overrun: byte {bad} behind the workspace of {key[2:]} was overwritten")


def pooled_supported(model, B, K):
    """True if the pooled kernels cover this table shape / batch (else the general kernels are used)."""
    return bool(_hip.lib().mkb_pool_supported(model._tables(), B, K))


class _PoolScoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ent, rel, modulus, model, sample, info, mode):
        B, K = sample.shape[0], info.size
        S = torch.empty((B, 2 * K), dtype=torch.float32, device=ent.device)
        ws = _workspace(model, B, K)
        with _hip.on_device(ent.device):
            _hip.check(_hip.lib().mkb_pool_score_fwd(model._tables(ent, rel, modulus), _hip.ptr(sample), _hip.ptr(info.pool),
                                                     _hip.ptr(info.cnt), B, K, mode, _hip.ptr(S), _hip.ptr(ws),
                                                     _hip.stream_ptr()), "mkb_pool_score_fwd")
        ctx.model, ctx.info, ctx.mode = model, info, mode
        ctx.save_for_backward(ent, rel, modulus, sample)
        return S.gather(1, info.pos.long())

    @staticmethod
    def backward(ctx, dneg):
        ent, rel, modulus, sample = ctx.saved_tensors
        model, info = ctx.model, ctx.info
        B, K = sample.shape[0], info.size
        G = torch.zeros((B, 2 * K), dtype=torch.float32, device=ent.device)
        G.scatter_add_(1, info.pos.long(), _hip.contiguous(dneg, torch.float32))
        # (a row-lazy optimizer's table takes its rows straight into .grad: _gradshare.direct)
        g_ent = _gradshare.direct(model.entity_embedding, ent, lambda: info.touched if info.touched is not None
                                  else torch.cat([info.pool, sample[:, 0::2].reshape(-1)]))
        fresh_e = False
        if g_ent is None:
            g_ent, fresh_e = _gradshare.take(model.entity_embedding, ent)  # (shared with the positive scores' backward of this pass)
        g_rel, fresh_r = _gradshare.take(model.relation_embedding, rel)
        g_mod = torch.zeros_like(modulus) if model.name == "pRotatE" else None
        gr = _hip.Grads(g_ent.data_ptr(), g_rel.data_ptr(), None if g_mod is None else g_mod.data_ptr())
        ws = _workspace(model, B, K)
        with _hip.on_device(ent.device):
            _hip.check(_hip.lib().mkb_pool_score_bwd(model._tables(ent, rel, modulus), gr, _hip.ptr(sample),
                                                     _hip.ptr(info.pool), _hip.ptr(info.cnt), B, K, ctx.mode, _hip.ptr(G),
                                                     _hip.ptr(ws), _hip.stream_ptr()), "mkb_pool_score_bwd")
        return (g_ent if fresh_e else None), (g_rel if fresh_r else None), g_mod, None, None, None, None


def pooled_forward(model, sample, info, mode_id):
    modulus = getattr(model, "modulus", None)
    if modulus is None:
        modulus = model.gamma
    return _PoolScoreFn.apply(model.entity_embedding, model.relation_embedding, modulus, model, sample, info, mode_id)


PoolInfo.enabled = True


class FusedTrainStep:
    """``loss = step(sample, weight, negative_sample, mode)``: fills ``param.grad`` (dense, accumulated) and returns
    the loss as a 0-dim device tensor.  ``positive_score`` [B,1] and ``negative_score`` [B,size] of the last call
    stay available for inspection."""

    def __init__(self, model, alpha):
        self.model, self.alpha = model, float(alpha)
        self._grads = None

    def _grad_buffers(self):
        m = self.model
        params = [m.entity_embedding, m.relation_embedding] + ([m.modulus] if m.name == "pRotatE" else [])
        for p in params:
            if p.grad is None:  # first step, or the optimizer's zero_grad(set_to_none=True)
                p.grad = torch.zeros_like(p)
        return _hip.Grads(m.entity_embedding.grad.data_ptr(), m.relation_embedding.grad.data_ptr(),
                          m.modulus.grad.data_ptr() if m.name == "pRotatE" else None)

    def sampled(self, sample, weight, sampler, mode, weight_sum=None):
        """``step(sample, weight, sampler.generate(sample, mode), mode)`` with the sampler folded into the optimizer's
        catch-up launch when the entity table steps row-lazily (``mkb_amd.optim.Adam(lazy_rows=True)``): no sampler
        launch, identical negatives.  The negatives of the call stay available as ``self.negative_sample``."""
        ent = self.model.entity_embedding
        lazy = _links.owner(ent)
        sample = _hip.contiguous(sample, torch.int64)
        if lazy is not None and sampler.size <= 512 and sample.is_cuda:
            neg = sampler.generate_with_catch_up(sample, mode, lazy, ent)
        else:
            neg = sampler.generate(sample=sample, mode=mode)
        self.negative_sample = neg
        return self(sample, weight, neg, mode, weight_sum=weight_sum)

    def __call__(self, sample, weight, negative_sample, mode, weight_sum=None):
        """``weight_sum``: optional device scalar = sum of weights of the WHOLE batch when these rows are one
        data-parallel shard of it (see mkb_amd.parallel); the returned loss is then this shard's share."""
        m = self.model
        info = getattr(negative_sample, "_mkb_pool", None)
        if info is None:
            raise ValueError("negative_sample does not come from mkb_amd.sampling.NegativeSampling.generate")
        mode_id = _hip.mode_id(mode)
        sample = _hip.contiguous(sample, torch.int64)
        weight = _hip.contiguous(weight, torch.float32)
        _hip.require_device(m.entity_embedding, sample, weight)
        B, K = sample.shape[0], info.size
        dev = sample.device
        pos = torch.empty((B, 1), dtype=torch.float32, device=dev)
        S = torch.empty((B, 2 * K), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws = _workspace(m, B, K)
        gr = self._grad_buffers()
        ent = m.entity_embedding
        lazy = _links.owner(ent)
        if lazy is not None:  # row-lazy Adam: the rows this step reads must be current before the forward pass
            ids = info.touched if info.touched is not None else torch.cat([info.pool, sample[:, 0], sample[:, 2]])
            st = lazy._state(ent)
            done = st.get("caught_up")
            if done is None or done[0] is not ids or done[1] != st["n"]:  # (sampled() already did it)
                lazy.catch_up(ent, ids)
            # Deferred real step: that launch consumed AND cleared the gradient row of every row it visited (= every entity row
            # this step writes), so unless an earlier backward of this same optimizer step has written since, those rows are
            # all-zero: the row kernels may store instead of read-modify-write (mkb_grads_t.rows_clear)
            # (autograd_wrote: an autograd backward of this same optimizer step has accumulated into .grad without marking rows)
            if (st.get("defer") and st["n"] >= 1 and _links.touched(ent) is None and not _links.autograd_wrote(ent)
                    and st.get("g") is not None and st["g"].data_ptr() == ent.grad.data_ptr()):
                gr.rows_clear = 1
            _links.mark_touched(ent, ids)  # accumulates when several steps share one optimizer.step()
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_pool_step(m._tables(), gr, _hip.ptr(sample), _hip.ptr(weight), _hip.ptr(info.pool),
                                                _hip.ptr(info.cnt), B, K, mode_id, self.alpha, _hip.ptr(weight_sum),
                                                _hip.ptr(pos), _hip.ptr(S),
                                                _hip.ptr(loss), _hip.ptr(ws), _hip.stream_ptr()), "mkb_pool_step")
        self.positive_score, self._S, self._info = pos, S, info
        return loss.reshape(())

    @property
    def negative_score(self):
        return self._S.gather(1, self._info.pos.long())

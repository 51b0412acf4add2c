"""``Adversarial`` -- self-adversarial negative-sampling loss (reference mkb/losses/adversarial.py:8-30).

``Adversarial(alpha)(positive_score, negative_score, weight)`` returns the scalar loss, differentiable w.r.t.
both score tensors.  Forward and the gradient seed come from ONE kernel sequence (``mkb_adversarial``): the
softmax weights are detached in the reference, so d loss / d scores is available in closed form as soon as the
row statistics are known; backward just scales the saved seeds by the upstream gradient.
"""
import torch

from .. import _hip

__all__ = ["Adversarial"]


class _AdversarialFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, neg, weight, alpha):
        B, K = neg.shape
        dev = neg.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dpos = torch.empty(B, dtype=torch.float32, device=dev)
        dneg = torch.empty((B, K), dtype=torch.float32, device=dev)
        scratch = torch.empty(B + 1, dtype=torch.float32, device=dev)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_adversarial(_hip.ptr(pos), _hip.ptr(neg), _hip.ptr(weight), None, B, K, alpha,
                                                  None, _hip.ptr(loss), _hip.ptr(dpos), _hip.ptr(dneg), _hip.ptr(scratch),
                                                  _hip.stream_ptr()), "mkb_adversarial")
        ctx.save_for_backward(dpos, dneg)
        ctx.pos_shape = None
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dpos, dneg = ctx.saved_tensors
        return (g * dpos).view(-1, 1), g * dneg, None, None


class Adversarial:
    def __init__(self, alpha=0.5):
        self.alpha = alpha

    def __call__(self, positive_score, negative_score, weight):
        _hip.require_device(positive_score, negative_score, weight)
        if positive_score.dim() != 2 or positive_score.size(1) != 1:
            raise ValueError("positive_score must be [B, 1]")  # the reference's squeeze(dim=1) contract
        pos = _hip.contiguous(positive_score, torch.float32)
        neg = _hip.contiguous(negative_score, torch.float32)
        w = _hip.contiguous(weight, torch.float32)
        if pos.shape[0] == 0:  # the reference divides two empty sums by W = 0 (losses/adversarial.py:28-30): nan, no launch
            return (pos.sum() + neg.sum() + w.sum()) / w.sum()
        return _AdversarialFn.apply(pos, neg, w, float(self.alpha))

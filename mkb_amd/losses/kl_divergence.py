"""``KlDivergence`` -- the distillation loss (reference mkb/losses/kl_divergence.py:8-29).

``KlDivergence()(student_score, teacher_score, T=1)``: mean over all entries of ``kl_div(log_softmax(student / T, dim=1),
softmax(teacher / T, dim=1), reduction="none")``.  Scores are the ``[n distributions, candidates]`` matrices a model
returns for the 3-D samples of ``distillation.Distillation.distill``.  Forward and both gradient seeds come from one
kernel sequence (``mkb_kl_divergence``); backward scales the saved seeds by the upstream gradient.
"""
import torch

from .. import _hip

__all__ = ["KlDivergence"]


class _KlFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, T):
        n, m = student.shape
        dev = student.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ds = torch.empty((n, m), dtype=torch.float32, device=dev)
        dt = torch.empty((n, m), dtype=torch.float32, device=dev) if teacher.requires_grad else None
        scratch = torch.empty(n, dtype=torch.float32, device=dev)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_kl_divergence(_hip.ptr(student), _hip.ptr(teacher), n, m, T, _hip.ptr(loss), _hip.ptr(ds),
                                                    _hip.ptr(dt), _hip.ptr(scratch), _hip.stream_ptr()), "mkb_kl_divergence")
        ctx.save_for_backward(ds, dt)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        ds, dt = ctx.saved_tensors
        return g * ds, None if dt is None else g * dt, None


class KlDivergence:
    def __init__(self):
        pass

    def __call__(self, student_score, teacher_score, T=1):
        _hip.require_device(student_score, teacher_score)
        if student_score.dim() != 2 or student_score.shape != teacher_score.shape:
            raise ValueError("student_score and teacher_score must be matrices of the same shape [distributions, candidates]")
        return _KlFn.apply(_hip.contiguous(student_score, torch.float32), _hip.contiguous(teacher_score, torch.float32), float(T))

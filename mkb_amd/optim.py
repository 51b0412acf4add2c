"""``mkb_amd.optim.Adam`` -- dense Adam with the element-wise semantics of ``torch.optim.Adam`` (the optimizer
the reference's training loops construct, README.md:123-126; stepped at compose/pipeline.py:238-240), running as
one HBM-streaming HIP kernel per parameter (``mkb_adam_step``) with ``optimizer.zero_grad()`` fused into it.

Opt-in: ``torch.optim.Adam`` keeps working with ``mkb_amd`` models (gradients are ordinary dense ``.grad``
tensors).  Dense means dense: every row of the tables moves every step through its moments, exactly like the
reference -- rows with a zero gradient still decay ``exp_avg`` / ``exp_avg_sq`` and step along the stale
momentum.
"""
import torch

from . import _hip

__all__ = ["Adam"]


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}
        self.step_count = 0
        self._zeroed = False

    def step(self):
        self.step_count += 1
        lib = _hip.lib()
        for p in self.params:
            if p.grad is None:
                continue
            _hip.require_device(p)
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p), [0])
            m, v, n = st
            n[0] += 1
            g = p.grad
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            with torch.cuda.device(p.device):
                _hip.check(lib.mkb_adam_step(_hip.ptr(p.data), _hip.ptr(g), _hip.ptr(m), _hip.ptr(v), p.numel(), n[0],
                                             self.lr, self.betas[0], self.betas[1], self.eps, 1, _hip.stream_ptr()),
                           "mkb_adam_step")
        self._zeroed = True

    def zero_grad(self, set_to_none=False):
        """Gradients were already cleared inside ``step`` (fused); they stay allocated so the next backward
        accumulates in place."""
        if self._zeroed and not set_to_none:
            self._zeroed = False
            return
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
        self._zeroed = False

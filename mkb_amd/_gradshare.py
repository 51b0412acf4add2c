"""Where the backward functions of ``model(...)`` put their table gradients.

The reference's step (README.md:448-474, compose/pipeline.py:211-236) scores the positive triples and the negatives with two
``model(...)`` calls, so autograd runs two of our backward functions against the same tables.

``take``: every backward function gets its OWN zero-filled dense buffer and returns it to autograd, which adds the contributions
of a pass.  (Round 5 let the functions of one pass share a buffer -- the first returned it, the others added their rows into
that very tensor and returned ``None``.  That mutates a tensor after handing it to autograd: with any OTHER contributor to the
table in the graph -- an L2 / L3 regulariser on ``entity_embedding``, a direct use of the Parameter -- the engine forms
``regulariser + buffer`` as a new tensor as soon as the first function returns, and rows added to the old buffer afterwards
never reach ``.grad``.  Whether the score functions are the sole contributors cannot be known from inside them, so nothing is
shared; tests/test_gpu_general.py::test_table_gradients_with_a_regulariser_on_both_tables.)

``direct``: a table stepped by a row-lazy ``mkb_amd.optim.Adam`` takes its rows straight into ``param.grad`` and autograd gets
``None`` -- no tensor is mutated behind autograd's back (what autograd itself accumulates into ``.grad`` is seen through the
tensor's version counter, ``_links.autograd_wrote``).
"""
import torch

from . import _links


def take(param, tensor):
    """-> (zero-filled buffer to accumulate into, True): the caller RETURNS it to autograd."""
    return torch.zeros_like(tensor), True


def direct(param, tensor, ids_fn):
    """The table's own ``.grad`` when a row-lazy ``mkb_amd.optim.Adam`` steps it, else ``None``.  With such an optimizer the
    backward functions add their rows STRAIGHT into ``param.grad`` (allocated by the optimizer, all-zero outside the rows that
    are pending), record the rows (``ids_fn() -> int64 ids``, duplicates allowed) and hand autograd ``None`` for the table --
    exactly what the fused step does.  Nothing dense is allocated, filled, added or stepped: the README loop
    (``model(sample)`` / ``model(sample, negatives, mode)`` / ``loss.backward()`` / ``optimizer.step()``) keeps the row-lazy
    route, where a dense buffer handed to autograd made the optimizer fall back to the dense kernel for good."""
    if param is None or _links.owner(param) is None or not _links.owner(param).direct_grads:
        return None
    g = param.grad
    if (g is None or tensor.data_ptr() != param.data_ptr() or g.shape != param.shape or not g.is_contiguous()
            or g.dtype != torch.float32 or g.device != param.device):
        return None
    # (rows a forward pass of THIS optimizer step count made current -- models/base.py:_make_current -- need no second visit
    # in front of the step launch)
    st = _links.owner(param)._state(param)
    _links.mark_touched(param, ids_fn(), current=st.get("fwd_n") == st["n"])
    _links.rebase(param)
    return g

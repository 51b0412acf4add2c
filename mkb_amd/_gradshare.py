"""One dense gradient buffer per parameter and backward pass.

The reference's step (README.md:448-474, compose/pipeline.py:211-236) scores the positive triples and the negatives with two
``model(...)`` calls, so autograd runs two of our backward functions against the same tables.  Each used to allocate its own
``zeros_like(table)`` (116 MB at the headline shape), and the autograd engine then ADDED the two dense tensors before handing
the sum to ``.grad``: a fill and a three-pass add of the whole table per step, for gradients that touch ~2,400 of 14,541 rows.
Here the first backward function of a pass allocates the buffer and returns it; the others of the same pass (same graph task,
same parameter) accumulate their rows into that very tensor -- our kernels add into their gradient buffers -- and return
``None`` for it, which autograd takes as "no contribution".  Outside a backward pass (no graph task) nothing is shared.
"""
import torch

from . import _links

_task = getattr(torch._C, "_current_graph_task_id", None)
_shared = _links.WeakIdTable()  # parameter -> (graph task id, buffer)


def take(param, tensor):
    """-> (buffer to accumulate into, fresh): ``fresh`` buffers are zero-filled and must be RETURNED to autograd by the caller;
    a shared one (``fresh`` False) was returned by an earlier backward function of this pass: return ``None`` instead."""
    tid = _task() if _task is not None else -1
    if tid < 0 or param is None or tensor.data_ptr() != param.data_ptr() or tensor.shape != param.shape:
        return torch.zeros_like(tensor), True
    entry = _shared.get(param)
    if entry is not None and entry[0] == tid and entry[1].shape == tensor.shape and entry[1].device == tensor.device:
        return entry[1], False
    buf = torch.zeros_like(tensor)
    _shared[param] = (tid, buf)
    return buf, True


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

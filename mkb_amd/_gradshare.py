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
from torch.utils.weak import WeakIdKeyDictionary

_task = getattr(torch._C, "_current_graph_task_id", None)
_shared = WeakIdKeyDictionary()  # parameter -> (graph task id, buffer)


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

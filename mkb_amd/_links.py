"""Side tables that tie a parameter to the ``mkb_amd.optim.Adam`` stepping it row-lazily and to the rows its pending step
has to visit.

They used to be attributes on the ``nn.Parameter``; torch pickles a Parameter's ``__dict__`` with it, so ``model.save()``
(models/base.py, the reference's mkb/models/base.py:41-46) serialised the whole optimizer -- moments, replay constants, a
ctypes sampler handle (which cannot be pickled at all) -- into the model file.  Identity-keyed weak tables keep the links
out of the model: nothing here survives or travels with a parameter.
"""
import torch
from torch.utils.weak import WeakIdKeyDictionary

__all__ = ["all_marks_current", "attach", "autograd_wrote", "clear_autograd_wrote", "detach", "mark_touched", "owner", "rebase", "take_touched", "touched"]

_owner = WeakIdKeyDictionary()    # parameter -> optimizer that defers its zero-gradient row steps
_touched = WeakIdKeyDictionary()  # parameter -> int64 ids of the rows written since the optimizer last stepped
_hooks = WeakIdKeyDictionary()    # parameter -> handle of the post-accumulate hook below
_wrote = WeakIdKeyDictionary()    # parameter -> True once AUTOGRAD has accumulated into .grad since the optimizer last stepped


_marks = WeakIdKeyDictionary()    # parameter -> [mark_touched calls, those of them whose rows a forward pass had made current] since the last step
_base = WeakIdKeyDictionary()     # parameter -> (data_ptr, version) of .grad when our own backward functions last looked at it


def _sig(p):
    g = p.grad
    return None if g is None else (g.data_ptr(), g._version)


def rebase(p):
    """Called by a backward function that adds its rows straight into ``p.grad`` (``_gradshare.direct``): whatever autograd
    accumulates into ``.grad`` behind it changes the tensor's version counter (or replaces the tensor)."""
    _base[p] = _sig(p)


def _note_autograd_write(p):
    # torch calls the hook at the end of every backward pass that reaches the parameter -- also when every backward function
    # handed autograd ``None`` for it; only a pass that really changed .grad counts
    if p not in _base or _base[p] != _sig(p):
        _wrote[p] = True


def autograd_wrote(p):
    """True if torch's autograd has accumulated into ``p.grad`` since the optimizer last stepped / cleared it (any route:
    ``model(sample, negatives, mode)`` + ``loss.backward()``, a regulariser on the table, ...).  The fused step bypasses autograd
    and does not count.  A gradient row written that way is NOT all-zero, so the fused step's row kernels must accumulate into
    it (``mkb_grads_t.rows_clear`` stays 0)."""
    return bool(_wrote.get(p, False))


def clear_autograd_wrote(p):
    _wrote.pop(p, None)


def owner(p):
    return _owner.get(p)


def attach(p, optimizer):
    _owner[p] = optimizer
    if p not in _hooks and hasattr(p, "register_post_accumulate_grad_hook"):
        _hooks[p] = p.register_post_accumulate_grad_hook(_note_autograd_write)


def detach(p):
    _owner.pop(p, None)
    _touched.pop(p, None)
    _wrote.pop(p, None)
    _base.pop(p, None)
    _marks.pop(p, None)
    h = _hooks.pop(p, None)
    if h is not None:
        h.remove()


def touched(p):
    return _touched.get(p)


def mark_touched(p, ids, replace=False, current=False):
    """Record the rows a backward pass wrote.  Several backward passes before one ``optimizer.step()`` accumulate
    (gradient accumulation): the lists are concatenated, so every written row takes the step and is cleared.
    ``replace=True``: ``ids`` already covers everything pending (e.g. the all-gathered union of a data-parallel step)."""
    prev = None if replace else _touched.get(p)
    _touched[p] = ids if prev is None or prev is ids else torch.cat([prev, ids])
    m = _marks.get(p)
    if m is None or replace:
        m = _marks[p] = [0, 0]
    m[0] += 1
    m[1] += 1 if current else 0


def all_marks_current(p):
    """True if every ``mark_touched`` since the last step said its rows had been made current through the optimizer's step
    count in front of the forward pass that read them (``current=True``): the step need not visit them again first."""
    m = _marks.get(p)
    return m is not None and m[0] > 0 and m[0] == m[1]


def take_touched(p):
    _wrote.pop(p, None)
    _marks.pop(p, None)
    return _touched.pop(p, None)

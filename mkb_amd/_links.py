"""Side tables that tie a parameter to the ``mkb_amd.optim.Adam`` stepping it row-lazily and to the rows its pending step
has to visit.

They used to be attributes on the ``nn.Parameter``; torch pickles a Parameter's ``__dict__`` with it, so ``model.save()``
(models/base.py, the reference's mkb/models/base.py:41-46) serialised the whole optimizer -- moments, replay constants, a
ctypes sampler handle (which cannot be pickled at all) -- into the model file.  Identity-keyed weak tables keep the links
out of the model: nothing here survives or travels with a parameter.
"""
import weakref

import torch


class WeakIdTable:
    """``parameter -> value``, keyed by identity, holding the parameter weakly.  (torch's ``WeakIdKeyDictionary`` does the
    same through a wrapper object whose ``__eq__`` / ``__hash__`` run in Python: ~1.5 us per lookup, 15 lookups per training step
    -- a tenth of the host's time per step at the small shapes.  Here: one ``dict.get(id(p))`` and one weak-reference call.)"""

    def __init__(self):
        self._d = {}

    def _entry(self, p):
        e = self._d.get(id(p))
        return e if e is not None and e[0]() is p else None

    def get(self, p, default=None):
        e = self._entry(p)
        return default if e is None else e[1]

    def __contains__(self, p):
        return self._entry(p) is not None

    def __getitem__(self, p):
        e = self._entry(p)
        if e is None:
            raise KeyError(p)
        return e[1]

    def __setitem__(self, p, value):
        e = self._entry(p)
        if e is not None:
            e[1] = value
            return
        key, d = id(p), self._d

        def gone(ref, key=key, d=d):
            cur = d.get(key)
            if cur is not None and cur[0] is ref:
                del d[key]

        d[key] = [weakref.ref(p, gone), value]

    def pop(self, p, default=None):
        e = self._entry(p)
        if e is None:
            return default
        del self._d[id(p)]
        return e[1]

    def __len__(self):
        return len(self._d)

__all__ = ["all_marks_current", "attach", "autograd_wrote", "clear_autograd_wrote", "detach", "mark_touched", "owner", "rebase", "take_touched", "touched"]

_owner = WeakIdTable()    # parameter -> optimizer that defers its zero-gradient row steps
_touched = WeakIdTable()  # parameter -> int64 ids of the rows written since the optimizer last stepped
_hooks = WeakIdTable()    # parameter -> handle of the post-accumulate hook below
_wrote = WeakIdTable()    # parameter -> True once AUTOGRAD has accumulated into .grad since the optimizer last stepped


_marks = WeakIdTable()    # parameter -> [mark_touched calls, those of them whose rows a forward pass had made current] since the last step
_base = WeakIdTable()     # parameter -> (data_ptr, version) of .grad when our own backward functions last looked at it


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

"""Side tables that tie a parameter to the ``mkb_amd.optim.Adam`` stepping it row-lazily and to the rows its pending step
has to visit.

They used to be attributes on the ``nn.Parameter``; torch pickles a Parameter's ``__dict__`` with it, so ``model.save()``
(models/base.py, the reference's mkb/models/base.py:41-46) serialised the whole optimizer -- moments, replay constants, a
ctypes sampler handle (which cannot be pickled at all) -- into the model file.  Identity-keyed weak tables keep the links
out of the model: nothing here survives or travels with a parameter.
"""
import torch
from torch.utils.weak import WeakIdKeyDictionary

__all__ = ["attach", "detach", "mark_touched", "owner", "take_touched", "touched"]

_owner = WeakIdKeyDictionary()    # parameter -> optimizer that defers its zero-gradient row steps
_touched = WeakIdKeyDictionary()  # parameter -> int64 ids of the rows written since the optimizer last stepped


def owner(p):
    return _owner.get(p)


def attach(p, optimizer):
    _owner[p] = optimizer


def detach(p):
    _owner.pop(p, None)
    _touched.pop(p, None)


def touched(p):
    return _touched.get(p)


def mark_touched(p, ids, replace=False):
    """Record the rows a backward pass wrote.  Several backward passes before one ``optimizer.step()`` accumulate
    (gradient accumulation): the lists are concatenated, so every written row takes the step and is cleared.
    ``replace=True``: ``ids`` already covers everything pending (e.g. the all-gathered union of a data-parallel step)."""
    prev = None if replace else _touched.get(p)
    _touched[p] = ids if prev is None or prev is ids else torch.cat([prev, ids])


def take_touched(p):
    return _touched.pop(p, None)

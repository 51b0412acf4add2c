"""Text layout shared by the ``__repr__`` of models and datasets: a title line followed by ``label  value`` rows, labels
right-aligned and values left-aligned to their longest entry (the layout the reference prints, e.g.
mkb/datasets/dataset.py:60-68, mkb/models/base.py:33-46)."""

__all__ = ["aligned_block"]


def aligned_block(title, fields):
    labels = [str(k) for k in fields]
    values = [str(v) for v in fields.values()]
    lw, vw = max(len(k) for k in labels), max(len(v) for v in values)
    rows = (f"{k:>{lw}}  {v:<{vw}}" for k, v in zip(labels, values))
    return "\n".join([title, *rows])

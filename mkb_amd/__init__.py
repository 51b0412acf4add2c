"""mkb_amd -- MI355X-native drop-in for the hot path of raphaelsty/mkb.

Same public names as the reference for the path it replaces::

    from mkb_amd import datasets, models, losses, sampling, compose, evaluation

* ``models.{TransE,RotatE,ComplEx,DistMult,pRotatE}``  forward/backward = hand-written gfx950 HIP kernels
* ``losses.Adversarial``                               fused softmax-weighted loss + gradient seed kernel
* ``sampling.NegativeSampling``                        bit-exact on-device MT19937 + filtered draw
* ``compose.Pipeline``                                 same training loop; fused step when it can
* ``evaluation.Evaluation``                            filtered ranking on device
* ``datasets``                                         host-side batch producer (same torch DataLoader order)
* ``distillation.{Distillation,KdmkbModel}`` + ``losses.KlDivergence``   second consumer of the scoring kernels
* ``table_rows`` / ``parallel``                        multi-GPU partitionings (row-sharded table, dims, batch rows)

All arithmetic runs in ``libmkb_hip.so`` (C ABI declared in ``include/mkb_hip.h``), loaded with ctypes by
``mkb_amd._hip``.  There is NO CPU compute fallback: tensors must live on a ROCm device and a missing
library raises at first use.
"""
__version__ = "0.1.0"

from . import compose, datasets, distillation, evaluation, fused, losses, models, optim, sampling, table_rows, utils  # noqa: F401

__all__ = ["compose", "datasets", "distillation", "evaluation", "fused", "losses", "models", "optim", "sampling", "table_rows", "utils"]

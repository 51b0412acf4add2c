from .adversarial import Adversarial
from .kl_divergence import KlDivergence

__all__ = ["Adversarial", "KlDivergence"]

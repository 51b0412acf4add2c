from .adversarial import Adversarial

__all__ = ["Adversarial"]

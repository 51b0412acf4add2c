from .distillation import Distillation
from .kdmkb_model import KdmkbModel
from .uniform_sampling import UniformSampling

__all__ = ["Distillation", "KdmkbModel", "UniformSampling"]

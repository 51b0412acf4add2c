from .base import BaseModel
from .complex import ComplEx
from .distmult import DistMult
from .protate import pRotatE
from .rotate import RotatE
from .transe import TransE

__all__ = ["BaseModel", "ComplEx", "DistMult", "RotatE", "TransE", "pRotatE"]

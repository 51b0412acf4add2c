"""ComplEx (reference mkb/models/complex.py:11-85): score = Re(<h, r, conj(t)>), rows = [real | imag]
(gamma unused)."""
from .base import BaseModel

__all__ = ["ComplEx"]


class ComplEx(BaseModel):
    def __init__(self, hidden_dim, entities, relations, gamma):
        super().__init__(hidden_dim=hidden_dim, relation_dim=hidden_dim * 2, entity_dim=hidden_dim * 2,
                         entities=entities, relations=relations, gamma=gamma)

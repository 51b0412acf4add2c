"""DistMult (reference mkb/models/distmult.py:9-75): score = sum_k h_k r_k t_k (gamma unused)."""
from .base import BaseModel

__all__ = ["DistMult"]


class DistMult(BaseModel):
    def __init__(self, hidden_dim, entities, relations, gamma):
        super().__init__(hidden_dim=hidden_dim, relation_dim=hidden_dim, entity_dim=hidden_dim, entities=entities,
                         relations=relations, gamma=gamma)

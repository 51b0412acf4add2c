"""TransE (reference mkb/models/transe.py:11-84): score = gamma - || h + r - t ||_1."""
from .base import BaseModel

__all__ = ["TransE"]


class TransE(BaseModel):
    def __init__(self, hidden_dim, entities, relations, gamma):
        super().__init__(hidden_dim=hidden_dim, relation_dim=hidden_dim, entity_dim=hidden_dim, entities=entities,
                         relations=relations, gamma=gamma)

"""pRotatE (reference mkb/models/protate.py:11-93): score = gamma - modulus * sum_k |sin((h + r - t)_k / (range/pi))|,
``modulus`` is trainable (protate.py:72, 91)."""
from math import pi

import torch
import torch.nn as nn

from .base import BaseModel

__all__ = ["pRotatE"]


class pRotatE(BaseModel):
    def __init__(self, hidden_dim, entities, relations, gamma):
        super().__init__(hidden_dim=hidden_dim, relation_dim=hidden_dim, entity_dim=hidden_dim, entities=entities,
                         relations=relations, gamma=gamma)
        self.pi = pi
        self.modulus = nn.Parameter(torch.Tensor([[0.5 * self.embedding_range.item()]]))

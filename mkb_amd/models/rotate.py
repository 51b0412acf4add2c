"""RotatE (reference mkb/models/rotate.py:11-99): score = gamma - sum_k | h_k e^{i phi_k} - t_k |,
phi = r / (embedding_range / pi); entity rows = [real | imag].  ``modulus`` exists (rotate.py:66-67) but is
unused by the score, so it never receives a gradient -- as in the reference."""
from math import pi

import torch
import torch.nn as nn

from .base import BaseModel

__all__ = ["RotatE"]


class RotatE(BaseModel):
    def __init__(self, hidden_dim, entities, relations, gamma):
        super().__init__(hidden_dim=hidden_dim, relation_dim=hidden_dim, entity_dim=hidden_dim * 2,
                         entities=entities, relations=relations, gamma=gamma)
        self.pi = pi
        self.modulus = nn.Parameter(torch.Tensor([[0.5 * self.embedding_range.item()]]))

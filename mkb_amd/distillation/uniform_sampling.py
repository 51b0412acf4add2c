"""``UniformSampling`` -- supervised uniform candidate sampling for distillation (reference
mkb/distillation/uniform_sampling.py:54-138): ONE draw of ``batch_size_entity`` shared entities and
``batch_size_relation`` shared relations per call (numpy ``RandomState.choice`` without replacement: the same stream as
the reference for the same seed), repeated for every positive triple; the ground truth later takes the last slot."""
import numpy as np
import torch

__all__ = ["UniformSampling"]


class UniformSampling:
    supervised = True  # the ground truth is part of every distribution
    depends_on_teacher = False  # nothing to refresh while the teacher trains (KdmkbModel.learn does not rebuild it)

    def __init__(self, batch_size_entity, batch_size_relation, seed=None, **kwargs):
        self.batch_size_entity = batch_size_entity
        self.batch_size_relation = batch_size_relation
        self._rng = np.random.RandomState(seed)

    def _draw(self, mapping, size):
        """-> (teacher ids, student ids) of ``size`` distinct shared items, as float rows like the reference's."""
        teacher = self._rng.choice(a=list(mapping.keys()), size=size, replace=False)
        student = [mapping[k] for k in teacher]
        return torch.Tensor(teacher).view(1, size), torch.Tensor(student).view(1, size)

    def get(self, mapping_entities, mapping_relations, positive_sample_size, **kwargs):
        ent_t, ent_s = self._draw(mapping_entities, self.batch_size_entity)      # entities first, then relations:
        rel_t, rel_s = self._draw(mapping_relations, self.batch_size_relation)   # the reference's RNG order
        rows = lambda x: x.repeat(positive_sample_size, 1)
        # head / relation / tail distributions of the teacher, then of the student (heads and tails share one draw)
        return rows(ent_t), rows(rel_t), rows(ent_t), rows(ent_s), rows(rel_s), rows(ent_s)

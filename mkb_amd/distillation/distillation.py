"""``Distillation`` -- knowledge distillation between two KGE models that share part of their entities / relations
(reference mkb/distillation/distillation.py:12-683; called from ``KdmkbModel.forward``, kdmkb_model.py:337-349).

For every positive triple of the teacher's batch whose parts both graphs know, three candidate lists are scored by teacher
and student -- P(head | r, t), P(relation | h, t), P(tail | h, r) -- and the student is pulled towards the teacher with
``losses.KlDivergence``.  The reference assembles the ``[n, candidates, 3]`` index tensors triple by triple on the host
(``.item()`` per element, deep copies, ``torch.stack`` of python lists); here they are built for the whole batch with a few
tensor operations on the device, from the same sampled candidates, and scored as 3-D samples (models/base.py:146-151) by
the HIP forward kernels (``mkb_score_fwd``).  Same numbers: tests pin the tensors and the loss to captures of the reference.
"""
import collections

import torch

from ..losses import KlDivergence

__all__ = ["Distillation"]


class Distillation:
    def __init__(self, teacher_entities, student_entities, teacher_relations, student_relations, sampling, device="cpu"):
        self.teacher_entities, self.student_entities = teacher_entities, student_entities
        self.teacher_relations, self.student_relations = teacher_relations, student_relations
        self.sampling, self.device = sampling, device
        shared = lambda t, s: collections.OrderedDict((i, s[label]) for label, i in t.items() if label in s)
        self.mapping_entities = shared(teacher_entities, student_entities)      # teacher id -> student id
        self.mapping_relations = shared(teacher_relations, student_relations)
        self._tables = {}

    # ------------------------------------------------------------------ which parts of a triple can be distilled
    def available(self, head, relation, tail):
        h, r, t = head in self.mapping_entities, relation in self.mapping_relations, tail in self.mapping_entities
        if self.sampling.supervised:  # the ground truth sits in every list: all three parts must be shared
            return dict.fromkeys(("head", "relation", "tail"), h and r and t)
        return {"head": r and t, "relation": h and t, "tail": h and r}

    def _lookup(self, device):
        """teacher id -> student id as dense device tables (-1 = not shared)."""
        tb = self._tables.get(device)
        if tb is None:
            def table(mapping, n):
                out = torch.full((n,), -1, dtype=torch.int64)
                if mapping:
                    out[torch.tensor(list(mapping.keys()))] = torch.tensor(list(mapping.values()))
                return out.to(device)
            tb = self._tables[device] = (table(self.mapping_entities, len(self.teacher_entities)),
                                         table(self.mapping_relations, len(self.teacher_relations)))
        return tb

    def distillation_tensors(self, sample, teacher=None):
        """-> {"head" | "relation" | "tail": (teacher [n, m, 3], student [n, m, 3])} int64 on ``sample``'s device, for the
        rows of ``sample`` (teacher ids) each part is available for; parts without rows are absent."""
        dev = sample.device
        ent_map, rel_map = self._lookup(dev)
        drawn = self.sampling.get(sample=sample, mapping_entities=self.mapping_entities, mapping_relations=self.mapping_relations,
                                  positive_sample_size=sample.shape[0], teacher=teacher)
        head_t, rel_t, tail_t, head_s, rel_s, tail_s = (d.to(device=dev, dtype=torch.int64) for d in drawn)
        h, r, t = sample[:, 0], sample[:, 1], sample[:, 2]
        hs, rs, ts = ent_map[h], rel_map[r], ent_map[t]
        ok_h, ok_r, ok_t = hs >= 0, rs >= 0, ts >= 0
        if self.sampling.supervised:
            every = ok_h & ok_r & ok_t
            rows = {"head": every, "relation": every, "tail": every}
        else:
            rows = {"head": ok_r & ok_t, "relation": ok_h & ok_t, "tail": ok_h & ok_r}
        out = {}
        for part, col, cand_t, cand_s, truth_t, truth_s in (("head", 0, head_t, head_s, h, hs), ("relation", 1, rel_t, rel_s, r, rs),
                                                            ("tail", 2, tail_t, tail_s, t, ts)):
            keep = rows[part]
            if not bool(keep.any()):
                continue
            ct, cs = cand_t[keep].clone(), cand_s[keep].clone()
            if self.sampling.supervised:  # the ground truth takes the last slot (distillation.py:313-314, 331-333)
                ct[:, -1], cs[:, -1] = truth_t[keep], truth_s[keep]
            m = ct.shape[1]
            fixed_t = torch.stack([h[keep], r[keep], t[keep]], dim=1).unsqueeze(1).expand(-1, m, -1).clone()
            fixed_s = torch.stack([hs[keep], rs[keep], ts[keep]], dim=1).unsqueeze(1).expand(-1, m, -1).clone()
            fixed_t[:, :, col], fixed_s[:, :, col] = ct, cs
            out[part] = (fixed_t, fixed_s)
        return out

    def distill(self, teacher, student, sample):
        """KL(teacher || student) summed over the three candidate lists; differentiable w.r.t. the student only."""
        dev = student.entity_embedding.device
        sample = sample.to(dev)
        loss = 0
        kl = KlDivergence()
        for teacher_x, student_x in self.distillation_tensors(sample, teacher=teacher).values():
            with torch.no_grad():
                teacher_score = teacher.distill(teacher_x.contiguous())
            loss = loss + kl(teacher_score=teacher_score, student_score=student.distill(student_x.contiguous()))
        return loss

"""``TopK``: best heads / relations / tails for a partial triple under a model -- the drop-in for ``mkb.utils.TopK``
(mkb/utils/top_k.py:7-234), a caller of the scoring path (``_get_rank``, top_k.py:227-234).  The candidate
triples are built on the model's device and scored in one ``model(sample)`` call; the order is the reference's
(descending ``argsort`` of the scores, first ``k``)."""
import torch

__all__ = ["TopK"]


class TopK:
    def __init__(self, entities, relations, device="cpu"):
        self.mapping_entities, self.mapping_relations = entities, relations
        self.reverse_mapping_entities = {i: label for label, i in entities.items()}
        self.reverse_mapping_relations = {i: label for label, i in relations.items()}
        self.entities = torch.tensor(list(entities.values()), dtype=torch.int64)
        self.relations = torch.tensor(list(relations.values()), dtype=torch.int64)
        self.device = device

    def _id(self, mapping, x):
        return mapping[x] if isinstance(x, str) else int(x)

    def _best(self, model, column, fixed, k):
        """Candidates vary in ``column`` (0 head, 1 relation, 2 tail); ``fixed`` = the two other ids in triple order."""
        dev = model.entity_embedding.device if str(self.device) != "cpu" or model.entity_embedding.is_cuda else self.device
        cand = (self.relations if column == 1 else self.entities).to(dev)
        cols = []
        it = iter(fixed)
        for c in range(3):
            cols.append(cand if c == column else torch.full_like(cand, next(it)))
        sample = torch.stack(cols, dim=1)
        training = model.training
        if training:
            model.eval()
        try:
            with torch.no_grad():
                rank = torch.argsort(model(sample), descending=True, dim=0).flatten()[:k]
        finally:
            if training:
                model.train()
        return cand[rank].tolist()

    def top_heads(self, k, model, relation, tail):
        ids = self._best(model, 0, (self._id(self.mapping_relations, relation), self._id(self.mapping_entities, tail)), k)
        return [self.reverse_mapping_entities[e] for e in ids]

    def top_relations(self, k, model, head, tail):
        ids = self._best(model, 1, (self._id(self.mapping_entities, head), self._id(self.mapping_entities, tail)), k)
        return [self.reverse_mapping_relations[r] for r in ids]

    def top_tails(self, k, model, head, relation):
        ids = self._best(model, 2, (self._id(self.mapping_entities, head), self._id(self.mapping_relations, relation)), k)
        return [self.reverse_mapping_entities[e] for e in ids]

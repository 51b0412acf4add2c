"""Scoring of given triples with a trained model: the drop-in for ``mkb.utils.FetchToPredict`` / ``make_prediction``
(mkb/utils/predict.py:9-106), a caller of the scoring path.  The batches are cut from one device tensor
(no DataLoader, no per-item ``__getitem__``); each is scored by ``model(x)`` (mode None -> ``mkb_score_fwd``)."""
import numpy as np
import torch

__all__ = ["FetchToPredict", "make_prediction"]


class FetchToPredict:
    """Iterates ``LongTensor [<= batch_size, 3]`` batches of ``dataset`` in order (predict.py:9-58).  ``num_workers`` is
    accepted for signature compatibility; ``device`` (extension) places the batches there directly."""

    def __init__(self, dataset, batch_size, num_workers=1, device=None):
        self.dataset, self.batch_size, self.num_workers, self.device = dataset, batch_size, num_workers, device

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        return torch.LongTensor(self.dataset[idx])

    def __iter__(self):
        if len(self.dataset) == 0:
            return
        triples = torch.as_tensor(np.asarray(self.dataset, dtype=np.int64).reshape(-1, 3))
        if self.device is not None:
            triples = triples.to(self.device)
        for lo in range(0, len(triples), self.batch_size):
            yield triples[lo: lo + self.batch_size]


def make_prediction(model, dataset, batch_size, num_workers=1, device="cuda"):
    """Scores of the triples of ``dataset`` under ``model``, flattened to ``[len(dataset)]`` (predict.py:61-106)."""
    with torch.no_grad():
        y_pred = [model(x) for x in FetchToPredict(dataset=dataset, batch_size=batch_size, num_workers=num_workers,
                                                    device=device)]
        return torch.cat(y_pred).flatten() if y_pred else torch.empty(0)

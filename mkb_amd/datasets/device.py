"""``DeviceBatches`` -- device-resident batch producer for ``compose.Pipeline``.

The reference's producer (mkb/datasets/dataset.py:188-194, 297-303) is a pair of torch ``DataLoader`` s with a worker
process, a python ``__getitem__`` per triple and an H2D copy per batch -- ~10 ms of host time per 1024-row batch,
10x the whole MI355X training step.  This keeps the training triples and their subsampling weights in HBM and
index-selects the batches on the device.  Same batch FORMAT (``{"sample", "weight", "mode"}``), same alternation
(head-batch then tail-batch, each view shuffled independently per epoch), same drop-nothing last batch; the
shuffle ORDER is torch's device ``randperm`` (seeded), not the reference's CPU ``RandomSampler`` order -- use the
plain ``Dataset`` when step-by-step parity with the reference is wanted (tests/test_gpu_pool.py does).
"""
import numpy as np
import torch

from .base import subsampling_weights

__all__ = ["DeviceBatches"]


class DeviceBatches:
    def __init__(self, dataset, device="cuda", seed=42):
        self.dataset = dataset
        self.device = torch.device(device)
        self.batch_size = dataset.batch_size
        self.shuffle = dataset.shuffle
        train = np.asarray(dataset.train, dtype=np.int64).reshape(-1, 3)
        self.triples = torch.as_tensor(train, device=self.device)
        self.weights = subsampling_weights(train).to(self.device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed if seed is not None else 0)

    def __getattr__(self, name):  # entities, relations, valid, test, train, n_entity, true_triples, ...
        return getattr(self.dataset, name)

    def __len__(self):
        n = -(-len(self.triples) // self.batch_size)
        return 2 * n

    def _order(self):
        n = len(self.triples)
        if self.shuffle:
            return torch.randperm(n, device=self.device, generator=self.gen)
        return torch.arange(n, device=self.device)

    def __iter__(self):
        views = []
        for mode in ("head-batch", "tail-batch"):  # one gather per view and epoch; the batches are slices of it
            order = self._order()
            views.append((self.triples[order], self.weights[order], mode))
        for lo in range(0, len(self.triples), self.batch_size):
            for triples, weights, mode in views:
                yield {"sample": triples[lo: lo + self.batch_size], "weight": weights[lo: lo + self.batch_size],
                       "mode": mode}

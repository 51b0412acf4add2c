"""Pooled (shared-candidate-pool) scoring path -- filled in with the pooled kernels."""


def pooled_forward(model, sample, pooled, mode_id):
    raise NotImplementedError

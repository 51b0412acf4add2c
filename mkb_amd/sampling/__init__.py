from .negative_sampling import NegativeSampling, positive_triples

__all__ = ["NegativeSampling", "positive_triples"]

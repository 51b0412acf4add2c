from .evaluation import Evaluation

__all__ = ["Evaluation"]

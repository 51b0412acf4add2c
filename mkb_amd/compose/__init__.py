from .pipeline import Pipeline

__all__ = ["Pipeline"]

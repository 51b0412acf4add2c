from .bar import Bar, BarRange
from .io import read_csv, read_json

__all__ = ["Bar", "BarRange", "read_csv", "read_json"]

from .bar import Bar, BarRange
from .io import read_csv, read_json
from .stats import Mean, RollingMean

__all__ = ["Bar", "BarRange", "Mean", "RollingMean", "read_csv", "read_json"]

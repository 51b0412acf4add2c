from .bar import Bar, BarRange
from .io import read_csv, read_json
from .predict import FetchToPredict, make_prediction
from .stats import Mean, RollingMean
from .top_k import TopK

__all__ = ["Bar", "BarRange", "FetchToPredict", "Mean", "RollingMean", "TopK", "make_prediction", "read_csv", "read_json"]

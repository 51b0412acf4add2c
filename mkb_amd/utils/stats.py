"""Running statistics with the interface of the two ``river.stats`` classes the reference uses
(compose/pipeline.py:189 ``RollingMean(1000)``; evaluation/evaluation.py:187-189 ``Mean()``)."""
import collections

__all__ = ["Mean", "RollingMean"]


class Mean:
    def __init__(self):
        self.n = 0
        self._mean = 0.0

    def update(self, x, w=1.0):
        self.n += w
        self._mean += (w / self.n) * (x - self._mean)
        return self

    def get(self):
        return self._mean


class RollingMean:
    def __init__(self, window_size):
        self.window_size = window_size
        self._w = collections.deque(maxlen=window_size)

    def update(self, x):
        self._w.append(x)
        return self

    def get(self):
        return sum(self._w) / len(self._w) if self._w else 0.0

"""tqdm wrappers with the reference's interface and refresh cadence (mkb/utils/bar.py:22-36, 55-69):
``set_description`` only touches the bar every ``update_every`` calls."""
import tqdm

__all__ = ["Bar", "BarRange"]


class Bar:
    def __init__(self, dataset, update_every, position=0):
        self.bar = tqdm.tqdm(dataset, position=position)
        self.update_every = update_every
        self.n = 0

    def __iter__(self):
        yield from self.bar

    def due(self):
        """True if the next ``set_description`` call will actually refresh the bar."""
        return self.n % self.update_every == 0

    def set_description(self, text):
        if self.n % self.update_every == 0:
            self.bar.set_description(text)
        self.n += 1


class BarRange(Bar):
    def __init__(self, step, update_every, position=0):
        super().__init__(range(step), update_every=update_every, position=position)

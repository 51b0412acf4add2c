"""Table generator shared by the CPU and GPU tests of the headline-shape slice fixture (no GPU imports here)."""
import numpy as np


def headline_tables(n_entity=14541, n_relation=237, hidden=1000, gamma=9.0, seed=2024):
    """Tables of tests/golden/headline_slice.npz: numpy's legacy generator, bit-stable across numpy versions and
    platforms -- the same draw tools/make_golden.py fed to the live reference (U(+-(gamma + 2) / hidden), models/base.py:81-100)."""
    rs = np.random.RandomState(seed)
    r = (gamma + 2.0) / hidden
    ent = rs.uniform(-r, r, size=(n_entity, 2 * hidden)).astype(np.float32)
    rel = rs.uniform(-r, r, size=(n_relation, hidden)).astype(np.float32)
    return ent, rel

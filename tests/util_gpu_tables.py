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


def eval_tables(name, n_entity=14541, n_relation=237, hidden=1000, gamma=9.0, seed=77):
    """Tables of tests/golden/eval_headline.npz for any of the five models (same legacy numpy draw, the model's own row lengths:
    transe.py:55-63, rotate.py:60-67, complex.py:55-63, distmult.py:53-61, protate.py:60-72) -> (ent, rel, modulus or None)."""
    de = 2 * hidden if name in ("RotatE", "ComplEx") else hidden
    dr = 2 * hidden if name == "ComplEx" else hidden
    rs = np.random.RandomState(seed)
    r = (gamma + 2.0) / hidden
    ent = rs.uniform(-r, r, size=(n_entity, de)).astype(np.float32)
    rel = rs.uniform(-r, r, size=(n_relation, dr)).astype(np.float32)
    modulus = np.array([[0.5 * np.float32(r)]], dtype=np.float32) if name in ("RotatE", "pRotatE") else None
    return ent, rel, modulus

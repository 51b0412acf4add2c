import ctypes
import json
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        p = GOLDEN / name
        if name.endswith(".json"):
            return json.loads(p.read_text())
        return np.load(p, allow_pickle=False)

    return load


@pytest.fixture(scope="session")
def liboracle():
    """The plain-C sampler restatement (oracle/Makefile)."""
    import subprocess

    so = ROOT / "oracle" / "liboracle.so"
    if not so.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")])
    return ctypes.CDLL(str(so))

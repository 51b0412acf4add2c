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


@pytest.fixture(autouse=True)
def _dirty_device_memory(request):
    """MKB_TEST_DIRTY_MEMORY=<GB>: before every -m gpu test, fill that much device memory with NaN and free it, so that the
    allocator hands the test dirty blocks instead of the zero pages of a fresh process (a kernel that reads bytes nobody wrote
    -- or runs past a buffer -- is silent on zero pages).  Off by default (it adds ~0.1 s per test)."""
    import os

    gb = os.environ.get("MKB_TEST_DIRTY_MEMORY")
    if gb and request.node.get_closest_marker("gpu") is not None:
        import torch

        if torch.cuda.is_available():
            blocks = [torch.full((256 << 20,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(max(1, int(float(gb))))]
            torch.cuda.synchronize()
            del blocks
    yield


@pytest.fixture(autouse=True)
def _workspace_guards(request):
    """MKB_WS_GUARD=1: after every -m gpu test, the pattern behind each cached pooled-kernel workspace must be intact
    (mkb_amd.fused.check_workspace_guards)."""
    yield
    import os

    if os.environ.get("MKB_WS_GUARD", "0") == "1" and request.node.get_closest_marker("gpu") is not None:
        from mkb_amd import fused

        fused.check_workspace_guards()

"""`bench.py --gpus 2` through its three partitionings, launched the way the driver launches it (torch.distributed.run,
one process per rank) but with both ranks on this one GPU over gloo (MKB_BENCH_ONE_DEVICE=1): a functional check of the
N > 1 entry path -- rank bookkeeping, the cross-rank consistency probe, the JSON line -- not a scaling number."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parallelism,scaling", [("dims", "weak"), ("rows", "weak"), ("table-rows", "weak"), ("rows", "strong"),
                                                 ("table-rows", "strong")])
def test_bench_two_ranks_on_one_device(parallelism, scaling):
    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MKB_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--config", "wn18rr-rotate", "--parallelism", parallelism, "--scaling", scaling]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["unit"] == "triples/s" and d["scaling"] == scaling
    assert d["config"]["global_batch"] == (2048 if scaling == "weak" else 1024)
    want = {"dims": "dims2", "rows": "dp2", "table-rows": "table-rows2"}[parallelism]
    assert d["config"]["parallelism"].startswith(want), d["config"]["parallelism"]

"""The row-sharded step (mkb_amd/table_rows.py, BASELINE config 5's partitioning) through RCCL on a ONE-GPU box.

Only 1-GPU boxes can be reached from the build container, and RCCL refuses two ranks on one device -- so the N > 1 tests of
tests/test_gpu_pool.py go through gloo.  Here the process group is `nccl` (= RCCL) with world size 1 and
MKB_ROWS_FORCE_COLLECTIVES=1 keeps the step from short-circuiting its collectives: all_to_all_single with split sizes (ids,
rows, gradient rows), the packed all-reduces (pool block + weight sum; pool-row gradients + relation gradient + loss), the
count exchange with its side-stream read-back, and the route planned one batch ahead on the side stream all run through the
RCCL backend, on the stream semantics (async_op + wait) the multi-GPU run uses.  Results must equal the single-process step.

Two forms: the collectives issued by libmkb_hip.so itself (mkb_rows_comm_*: its own RCCL communicators, grouped all-reduce +
send / recv on the step's stream, in-band split sizes, the default on an `nccl` group) and the round-4 form through
torch.distributed (MKB_ROWS_PY_COLLECTIVES=1, what gloo groups still use)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("form", ["library", "library-2-ahead", "torch"])
@pytest.mark.parametrize("name,hidden,K,size", [("RotatE", 500, 256, "yago"), ("RotatE", 40, 16, "big"), ("pRotatE", 20, 16, "small")])
def test_row_sharded_step_through_rccl_at_world_1(name, hidden, K, size, form):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "tr_worker.py"), name, str(hidden), str(K), size]
    env = dict(os.environ, MKB_TR_BACKEND="nccl", MKB_ROWS_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MKB_ROWS_PY_COLLECTIVES="1" if form == "torch" else "0", MKB_TR_LOOKAHEAD="2" if form.endswith("2-ahead") else "1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "TR_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert "nccl collectives_run True" in out.stdout, out.stdout[-500:]
    assert f"lib_collectives {form != 'torch'}" in out.stdout, out.stdout[-500:]

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
           "--config", "wn18rr-rotate", "--parallelism", parallelism, "--scaling", scaling, "--no-extras"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["unit"] == "triples/s" and d["scaling"] == scaling
    assert d["config"]["global_batch"] == (2048 if scaling == "weak" else 1024)
    want = {"dims": "dims2", "rows": "dp2", "table-rows": "table-rows2"}[parallelism]
    assert d["config"]["parallelism"].startswith(want), d["config"]["parallelism"]


def test_bench_default_partitioning_is_table_rows_and_reports_the_others():
    """What the driver launches (no --parallelism): the north_star partitioning is the line's value; dims and rows ride in
    the same line under other_partitionings."""
    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MKB_BENCH_ONE_DEVICE="1")
    env.pop("MKB_BENCH_PARALLELISM", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--config", "wn18rr-rotate"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["parallelism"].startswith("table-rows2") and "partitioning_failures" not in d
    assert set(d["other_partitionings"]) == {"dims", "rows"} and all(v["value"] > 0 for v in d["other_partitionings"].values())
    assert "extras_error" not in d, d["extras_error"]
    # ... and the other way round: another partitioning as the line's value, the row-sharded step among the extras (measured
    # without the kernel-class probe: its timed batches must follow the warm-up batches without a gap -- the step plans the next
    # batch's routes, a skipped batch is an error there; round 4 found the extras ending in exactly that error)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--config", "wn18rr-rotate", "--parallelism", "dims"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["parallelism"].startswith("dims2") and "extras_error" not in d, d.get("extras_error")
    assert set(d["other_partitionings"]) == {"table-rows", "rows"} and all(v["value"] > 0 for v in d["other_partitionings"].values())


def test_bench_table_rows_code_path_on_one_rank():
    """--force-parallelism: the row-sharded step at world 1 (no collective): what one rank computes per step."""
    from conftest import ROOT

    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--config", "yago310-rotate", "--parallelism",
           "table-rows", "--force-parallelism", "--no-traffic"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["parallelism"].startswith("table-rows1") and d["value"] > 0


@pytest.mark.parametrize("timeout", [None, "0.2"])
def test_bench_safety_line_of_the_row_sharded_step(timeout):
    """N > 1: a short measurement with the collectives in torch.distributed runs in front of the library-issued form and rides
    the line (``torch_distributed_form``); a watchdog prints it as the line when the main measurement does not finish in time
    (forced here with a timeout no measurement can meet).  One device, gloo: MKB_BENCH_FORCE_SAFETY=1 switches the path on."""
    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MKB_BENCH_ONE_DEVICE="1", MKB_BENCH_FORCE_SAFETY="1")
    env.pop("MKB_BENCH_PARALLELISM", None)
    if timeout:
        env["MKB_BENCH_MAIN_TIMEOUT"] = timeout
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "60" if timeout else "4", "--warmup", "2",
           "--config", "wn18rr-rotate", "--no-extras"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"].startswith("table-rows2")
    if timeout:
        assert "did not finish in time" in d["error"] and d["steps"] == 40
    else:
        assert "error" not in d and d["torch_distributed_form"]["value"] > 0 and d["steps"] == 4


def test_bench_line_reports_the_median_window_and_every_window():
    """Round 6: the timed region is R back-to-back windows of --steps steps; ms_per_step / value are the MEDIAN window (an actual
    window: the lower median for an even count) and nothing is dropped: every window is listed with its instruments."""
    from conftest import ROOT

    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--windows", "6", "--config", "wn18rr-rotate", "--no-traffic"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["steps"] == 6 and d["warmup"] == 2 and d["windows"] == 6 and len(d["windows_ms"]) == 6 == len(d["windows_detail"])
    per_step = sorted(w / 6 for w in d["windows_ms"])
    assert abs(d["ms_per_step"] - per_step[2]) < 1e-6 * per_step[2] + 1e-5, (d["ms_per_step"], per_step)  # lower median of six
    assert abs(d["value"] - 1024 * 129 / (d["ms_per_step"] / 1e3)) < 1e-3 * d["value"]                   # B * (K + 1) per step
    for w in d["windows_detail"]:
        assert {"ms", "host_enqueue_ms", "device_ms", "largest_submit_gap_ms", "largest_submit_gap_at_step", "gc_ms"} <= set(w)
        assert 0 < w["host_enqueue_ms"] <= w["ms"] + 1e-3 and 0 < w["device_ms"] <= w["ms"] + 0.5
    assert d["t_host_ms_per_step"] > 0 and isinstance(d["gc_collections_during_windows"], list)

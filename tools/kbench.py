"""Kernel-level A/B harness: build libmkb_hip variants with -D flags (in the build container), then time the
per-kernel-class durations of the headline training step for each variant on the GPU.

    python tools/kbench.py build name1:-DFLAG1,-DFLAG2 name2:...     (container; writes gpurun_variants/*.so)
    python tools/kbench.py run [name ...]                              (GPU box; prints a table)
"""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
VDIR = ROOT / "variants"


def build(specs):
    sys.path.insert(0, str(ROOT))
    from mkb_amd.csrc import build as hb

    VDIR.mkdir(exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition(":")
        flags = [f for f in flags.split(",") if f]
        hb.build(force=True, verbose=False, extra_flags=flags, out=VDIR / f"lib_{name}.so", objdir=VDIR / f"obj_{name}")
        print("built", name, flags)


def run_one(steps=40):
    sys.path.insert(0, str(ROOT))
    import torch

    import bench
    from mkb_amd import _hip

    ctx = bench.build(torch.device("cuda", 0), 0, 1)
    kinds = list(_hip.PROF_KINDS)
    for i in range(10):
        bench.run_step(ctx, i)
    torch.cuda.synchronize()
    for k in kinds:
        _hip.profile_enable(k, True)
    import time
    t0 = time.perf_counter()
    for i in range(steps):
        bench.run_step(ctx, 10 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"ms_per_step": round(dt * 1e3, 4)}
    for k in kinds:
        n, ms = _hip.profile_read(k)
        res[k] = round(ms / steps * 1e3, 1)  # us per step
    print(json.dumps(res))


def run(names):
    if not names:
        names = sorted(p.stem[4:] for p in VDIR.glob("lib_*.so"))
    for name in names:
        env = dict(os.environ, MKB_HIP_LIB=str(VDIR / f"lib_{name}.so"))
        out = subprocess.run([sys.executable, __file__, "_one"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(f"{name:28s}", line[-1] if line else ("FAILED " + out.stderr[-400:]))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        build(sys.argv[2:])
    elif cmd == "run":
        run(sys.argv[2:])
    elif cmd == "_one":
        run_one()

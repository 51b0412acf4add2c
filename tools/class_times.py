"""Per-kernel-class times of one configuration's step (HIP events inside the library, mkb_profile_*), head- and tail-batch
steps apart:   python tools/class_times.py <config> [steps]        (GPU box; MKB_BENCH_NO_RIDE=1 etc. apply)"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from mkb_amd import _hip  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bench.__dict__.update(bench.CONFIGS[cfg])
ctx = bench.build(torch.device("cuda", 0), 0, 1)
ctx["rows_per_rank"] = bench.B
import os
W = int(os.environ.get("CLASS_TIMES_WARMUP", "20"))  # (an odd count flips head / tail against the optimizer's step parity)
for i in range(20):
    bench.run_step(ctx, i + (W - 20))
torch.cuda.synchronize()
kinds = list(_hip.PROF_KINDS)
for parity, name in ((0, "head-batch"), (1, "tail-batch")):
    tot = {k: 0.0 for k in kinds}
    for i in range(steps):
        on = (W + i) % 2 == parity
        for k in kinds:
            _hip.profile_enable(k, on)
        bench.run_step(ctx, W + i)
        if on:
            torch.cuda.synchronize()
            for k in kinds:
                tot[k] += _hip.profile_read(k)[1]
    print(name, json.dumps({k: round(v / (steps / 2) * 1e3, 1) for k, v in tot.items() if v}))

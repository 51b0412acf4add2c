#!/bin/bash
O=gpurun_out/j; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --windows 10 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants"
$B > $O/plain.json 2>/dev/null
MKB_BENCH_PREWARM_OTHER=300 $B > $O/other300.json 2>/dev/null
python - <<'PY'
import json
for n in ("plain","other300"):
    d=json.loads(open(f"gpurun_out/j/{n}.json").read().strip().splitlines()[-1])
    print(n, d["windows_ms_per_step"])
PY

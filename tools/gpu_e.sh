#!/bin/bash
mkdir -p gpurun_out/e
B="python bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants"
$B > gpurun_out/e/cold.json 2>/dev/null
$B --preheat-ms 300 > gpurun_out/e/preheat300.json 2>/dev/null
$B --preheat-ms 1500 > gpurun_out/e/preheat1500.json 2>/dev/null
python - <<'PY'
import json
for n in ("cold","preheat300","preheat1500"):
    d=json.loads(open(f"gpurun_out/e/{n}.json").read().strip().splitlines()[-1])
    print(n, d["windows_ms_per_step"])
PY
python -m pytest tests -q -m gpu > gpurun_out/e/suite.log 2>&1; tail -5 gpurun_out/e/suite.log

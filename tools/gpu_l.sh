#!/bin/bash
O=gpurun_out/l; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --windows 10 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants"
MKB_BENCH_NO_DRAW_AHEAD=1 MKB_BENCH_FIXED_BATCH=1 $B > $O/fixed.json 2>$O/fixed.err
MKB_BENCH_NO_DRAW_AHEAD=1 MKB_BENCH_NO_RIDE=1 $B > $O/noride.json 2>$O/noride.err
python - <<'PY'
import json
for n in ("fixed","noride"):
    try:
        d=json.loads(open(f"gpurun_out/l/{n}.json").read().strip().splitlines()[-1])
        print(n, d["windows_ms_per_step"])
    except Exception as e: print(n, "ERR", e, open(f"gpurun_out/l/{n}.err").read()[-600:])
PY

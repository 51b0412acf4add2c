#!/bin/bash
O=gpurun_out/m; mkdir -p $O
MKB_BENCH_SCLK=1 python bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants > $O/sclk.json 2>$O/sclk.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/m/sclk.json").read().strip().splitlines()[-1])
print([ (round(w["ms"]/20,4), round(w["sclk_mhz_after"])) for w in d["windows_detail"]])
PY
tail -3 $O/sclk.err

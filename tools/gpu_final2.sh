#!/bin/bash
O=gpurun_out/final2; mkdir -p $O
python -m pytest tests -q -m gpu > $O/suite.log 2>&1; tail -3 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2> $O/driver.err; python -c "
import json; d=json.loads(open('$O/driver.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,4), d['windows_ms_per_step'], d['roofline']['avg_kernel_us'], d['roofline']['frac'], d['roofline']['traffic'])"

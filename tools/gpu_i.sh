#!/bin/bash
R=$(pwd); O=$R/gpurun_out/i; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in wn18rr-rotate fb15k237-transe fb15k237-complex yago310-rotate; do
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python $R/bench.py --config $c --steps 40 --warmup 5 --windows 8 --no-traffic --profile-kernel none > $O/bench_$c.json 2> $O/kt.log
echo "== $c $(python -c "import json;d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['windows_ms_per_step'])")"
python $R/tools/warmup_trend.py $(find $O/kt -name "*.db" | head -1) 40
rm -rf $O/kt
done 2>&1 | tee $O/trend.txt

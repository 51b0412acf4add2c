#!/bin/bash
R=$(pwd); O=$R/gpurun_out/f; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o run -- python $R/bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants --profile-kernel none > $O/bench.json 2> $O/kt.log
cd $R
python tools/warmup_trend.py $(find $O/kt -name "*.db" | head -1) 20 | tee $O/trend.txt
rm -rf $O/kt

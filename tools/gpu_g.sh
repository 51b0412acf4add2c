#!/bin/bash
R=$(pwd); O=$R/gpurun_out/g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for lr in 0 5e-5 5e-4; do
MKB_BENCH_LR=$lr timeout 300 rocprofv3 --kernel-trace -d $O/kt$lr -o run -- python $R/bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants --profile-kernel none > $O/bench$lr.json 2> $O/kt.log
echo "== lr $lr"; python $R/tools/warmup_trend.py $(find $O/kt$lr -name "*.db" | head -1) 20 | head -5
rm -rf $O/kt$lr
done 2>&1 | tee $O/trend.txt

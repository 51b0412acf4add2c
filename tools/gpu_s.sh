#!/bin/bash
R=$(pwd); O=$R/gpurun_out/s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in base skipdx; do
MKB_HIP_LIB=$R/variants/lib_$v.so timeout 300 rocprofv3 --kernel-trace -d $O/kt_$v -o run -- python $R/bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants --profile-kernel none > $O/bench_$v.json 2> $O/kt.log
echo "== $v"; python $R/tools/warmup_trend.py $(find $O/kt_$v -name "*.db" | head -1) 20 | grep "bwd1"
rm -rf $O/kt_$v
done; done 2>&1 | tee $O/trend.txt

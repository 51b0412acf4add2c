#!/bin/bash
R=$(pwd); O=$R/gpurun_out/t; mkdir -p $O
python -m pytest tests/test_gpu_dense_pass.py tests/test_gpu_pool.py -q -m gpu -x -k "dense or every_row or headline or config2 or full_size" > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for v in base rider; do
MKB_HIP_LIB=$R/variants/lib_$v.so timeout 300 rocprofv3 --kernel-trace -d $O/kt_$v -o run -- python $R/bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants --profile-kernel none > $O/bench_$v.json 2> $O/kt.log
echo "== $v $(python -c "import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print(d['windows_ms_per_step'][-5:], d['loss'])")"; python $R/tools/warmup_trend.py $(find $O/kt_$v -name "*.db" | head -1) 20 | grep "bwd"
rm -rf $O/kt_$v
done; done 2>&1 | tee $O/trend.txt
cd $R
for v in base rider; do MKB_HIP_LIB=$R/variants/lib_$v.so python bench.py --steps 20 --warmup 5 --windows 15 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', round(d['ms_per_step'],4), d['loss'])"; done | tee -a $O/trend.txt

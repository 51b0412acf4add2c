#!/bin/bash
# round 5: A/B through bench.py (env settings per line)
R=$(pwd); O=$R/gpurun_out/r5_ab_small; rm -rf $O; mkdir -p $O
run() { c=$1; shift; echo "== $c $*" >> $O/ab.txt
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 --no-traffic --steps 300 --warmup 30 --profile-kernel ${PK:-auto} 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print(round(j['ms_per_step'],4), (r.get('kernel') or '')[:70], r.get('avg_kernel_us'), r.get('frac'), j.get('loss'))" >> $O/ab.txt 2>&1; }
export PK=pool_bwd_q
run fb15k237-transe X=1
run fb15k237-transe MKB_POOL_DENSE=1
run fb15k237-transe X=1
run fb15k237-transe MKB_POOL_DENSE=1
cat $O/ab.txt

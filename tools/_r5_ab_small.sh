#!/bin/bash
# round 5: backward variants on the small shapes and on TransE: env settings A/B'd through bench.py
R=$(pwd); O=$R/gpurun_out/r5_ab_small; rm -rf $O; mkdir -p $O
run() { c=$1; shift; echo "== $c $*" >> $O/ab.txt
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 --no-traffic --steps 400 --warmup 40 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print(round(j['ms_per_step'],4), (r.get('kernel') or '')[:70], r.get('kernel_us'), r.get('frac'))" >> $O/ab.txt 2>&1; }
run umls-transe X=1
run umls-transe MKB_POOL_SMALL=0
run fb15k237-transe X=1
run fb15k237-transe MKB_POOL_BWD1=0
run fb15k237-transe MKB_POOL_BWD1_NO_K4=1
cat $O/ab.txt

#!/bin/bash
# round 5: A/B through bench.py (env settings per line)
R=$(pwd); O=$R/gpurun_out/r5_ab_small; rm -rf $O; mkdir -p $O
run() { c=$1; shift; echo "== $c $*" >> $O/ab.txt
  env "$@" timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 --no-traffic --steps 200 --warmup 20 --profile-kernel ${PK:-auto} 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print(round(j['ms_per_step'],4), (r.get('kernel') or '')[:70], r.get('avg_kernel_us'), r.get('frac'), j.get('loss'))" >> $O/ab.txt 2>&1; }
export PK=pool_fwd
run fb15k237-transe MKB_POOL_TILE_ONLY=d
run fb15k237-transe MKB_POOL_TILE_ONLY=f
run fb15k237-transe MKB_POOL_TILE_ONLY=d MKB_POOL_TILE_KS=4
run fb15k237-transe MKB_POOL_TILE_ONLY=d MKB_POOL_TILE_KS=16
run headline MKB_POOL_TILE_ONLY=d
run headline MKB_POOL_TILE_ONLY=f
cat $O/ab.txt

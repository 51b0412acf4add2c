#!/bin/bash
# A/B of the single-pass backward's dense pass on the GPU box: step time of four configs with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --mrr-epochs 0 --no-variants --no-traffic --steps 400 --warmup 40"
for c in headline wn18rr-rotate yago310-rotate fb15k237-transe; do
  for d in 0 1; do
    echo -n "$c MKB_POOL_DENSE=$d  "; MKB_POOL_DENSE=$d $B --config $c 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['roofline'].get('kernel'), j['roofline'].get('launch_us', j['roofline'].get('achieved')))"
  done
done > gpurun_out/ab_dense.txt 2>&1
cat gpurun_out/ab_dense.txt

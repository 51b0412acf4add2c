#!/bin/bash
# round 5: the two-pass merged backward (MKB_POOL_BWD1=0) against the single-pass kernel on the small shapes
R=$(pwd); O=$R/gpurun_out/r5_ab_small; rm -rf $O; mkdir -p $O
for c in umls-transe wn18rr-rotate; do
  for v in "" "MKB_POOL_BWD1=0"; do
    echo "== $c $v" >> $O/ab.txt
    env $v timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 --no-traffic --steps 400 --warmup 40 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel'][:60] if j.get('roofline') else None, j['roofline'].get('kernel_us') if j.get('roofline') else None)" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt

#!/bin/bash
# tools/ab_env.sh <config> <kernel class> "<ENV=.. ENV=..>" ["<ENV=..>" ...]: A/B of environment switches through bench.py on the GPU
# box, one line per setting: ms per step, the class's kernel, its mean duration under the bench's HIP events, frac, loss.
# (round 5's measurements: MKB_POOL_SMALL, MKB_POOL_TILE_ONLY, MKB_POOL_TILE_KS / _FSL, MKB_POOL_DENSE, MKB_POOL_BWD1 ...)
cfg=$1; cls=$2; shift 2
for v in "X=1" "$@"; do
  echo -n "$cfg [$v]: "
  env $v timeout 300 python bench.py --config $cfg --no-cpu-baseline --mrr-epochs 0 --no-traffic --steps 300 --warmup 30 --profile-kernel $cls 2>/dev/null | tail -1 |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print(round(j['ms_per_step'],4), (r.get('kernel') or '')[:60], r.get('avg_kernel_us') and round(r['avg_kernel_us'],1), r.get('frac') and round(r['frac'],3), j.get('loss'))"
done

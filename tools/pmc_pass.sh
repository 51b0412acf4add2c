#!/bin/bash
# tools/pmc_pass.sh <tag> <config> "<counters>" [ENV=..]: one rocprofv3 --pmc pass over a short bench run, per-kernel means
tag=$1; cfg=$2; ctr=$3; shift 3
R=$(pwd); O=$R/gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $O -o run -- python $R/bench.py --config $cfg --steps 12 --warmup 4 --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel none --no-traffic > $O/bench.json 2> $O/log.txt
cd $R
python tools/pmc_summary.py $(find $O -name "*.db" | head -1) | grep -E "pool_fwd|pool_bwd|kernel " | cut -c1-40,72-140
rm -rf $O

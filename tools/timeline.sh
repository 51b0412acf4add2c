#!/bin/bash
# tools/timeline.sh <tag> <config> [ENV=.. ...]: rocprofv3 kernel trace of a short bench run of <config> on the GPU box;
# writes gpurun_out/tl_<tag>.txt = one training step as a timeline + the per-kernel table of the mkb:: kernels, and prints it
tag=$1; cfg=$2; shift 2
R=$(pwd); O=$R/gpurun_out/tl_$tag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace -d $O -o run -- python $R/bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel none --no-traffic > $O/bench.json 2> $O/log.txt
cd $R
db=$(find $O -name "*.db" | head -1)
{ echo "# $cfg $*: $(python -c "import json,sys; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('ms_per_step under the profiler', round(j['ms_per_step'],4))" 2>/dev/null)"
  python tools/prof_summary.py $db timeline
  python tools/prof_summary.py $db | grep -E "^kernel|mkb::" | head -16; } > $R/gpurun_out/tl_$tag.txt 2>&1
rm -rf $O
cat $R/gpurun_out/tl_$tag.txt

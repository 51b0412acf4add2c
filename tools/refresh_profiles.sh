#!/bin/bash
# Runs on the GPU box (gpurun): regenerates the evidence under gpurun_out/ that is then copied into profiles/.
#   kernel trace (rocprofv3 --kernel-trace, summarised by tools/prof_summary.py), HBM traffic (two separate --pmc
#   passes, tools/pmc_summary.py), the default bench line, the bench line under the profiler, other configs.
set -u
R=$(pwd)
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --mrr-epochs 0"
rocprofv3 --kernel-trace -d $O/ktrace -o run -- $BENCH --steps 60 --warmup 10 > $O/bench_under_rocprof.json 2> $O/ktrace.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_$c.json 2> $O/pmc_$c.log
done
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/pmc_sq -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_sq.json 2> $O/pmc_sq.log
cd $R
python tools/prof_summary.py $(ls $O/ktrace/*.db | head -1) > $O/kernel_trace_stats.txt
python tools/prof_summary.py $(ls $O/ktrace/*.db | head -1) timeline > $O/kernel_timeline.txt
python tools/pmc_summary.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) > $O/pmc_hbm_traffic.txt
python tools/pmc_summary.py $(ls $O/pmc_sq/*.db | head -1) > $O/pmc_sq.txt 2>&1
python bench.py --breakdown > $O/bench_default.json 2> $O/bench_default.err
for c in wn18rr-rotate fb15k237-complex fb15k237-transe fb15k237-distmult yago310-rotate; do
  python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 2>/dev/null | tail -1 >> $O/bench_configs.jsonl
done
for n in 1 2 4 8; do echo "world=$n $(python tools/shard_emulate.py $n 2>/dev/null | tail -1)" >> $O/shard_emulate.txt; done
[ -x tools/ubench/valu_chain ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/ubench/valu_chain tools/ubench/valu_chain.hip
tools/ubench/valu_chain > $O/valu_ubench.txt 2>&1
rm -rf $O/ktrace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq   # keep the summaries, not the databases
ls -la $O

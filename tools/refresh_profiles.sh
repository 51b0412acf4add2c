#!/bin/bash
# Runs on the GPU box (gpurun): regenerates the evidence under gpurun_out/refresh/ that is then copied into profiles/.
#   kernel trace (rocprofv3 --kernel-trace, summarised by tools/prof_summary.py), HBM traffic (two separate --pmc
#   passes, tools/pmc_summary.py), SQ counters, the default bench line, the bench line under the profiler, other configs.
# Every profiler pass runs under `timeout`: a pass that wedges must not eat the GPU budget.
set -u
R=$(pwd)
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O   # (everything below is regenerated)
cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --mrr-epochs 0 --no-variants --no-traffic"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktrace -o run -- $BENCH --steps 60 --warmup 10 > $O/bench_under_rocprof.json 2> $O/ktrace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_$c.json 2> $O/pmc_$c.log
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $O/pmc_sq -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_sq.json 2> $O/pmc_sq.log
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_clk -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_clk.json 2> $O/pmc_clk.log
cd $R
python tools/prof_summary.py $(find $O/ktrace -name "*.db" | head -1) > $O/kernel_trace_stats.txt
python tools/prof_summary.py $(find $O/ktrace -name "*.db" | head -1) timeline > $O/kernel_timeline.txt
cp $(ls $O/ktrace/*kernel_stats.csv 2>/dev/null | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) > $O/pmc_hbm_traffic.txt
python tools/pmc_summary.py $(find $O/pmc_sq -name "*.db" | head -1) $(find $O/pmc_clk -name "*.db" | head -1) > $O/pmc_sq.txt 2>&1
timeout 900 python bench.py --breakdown > $O/bench_default.json 2> $O/bench_default.err
# the driver's protocol, three fresh processes
for i in 1 2 3; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_driver_protocol.jsonl; done
# which launch of the step gets faster over the first ~150 steps of a process (the trend inside windows_ms)
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/kt_trend -o run -- python $R/bench.py --steps 20 --warmup 5 --windows 12 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants --profile-kernel none > /dev/null 2> $O/kt_trend.log
cd $R
{ echo "# mean duration (us) of every kernel of the headline step by launch order, 20 launches per column (rocprofv3 --kernel-trace of"
  echo "# bench.py --steps 20 --warmup 5 --windows 12): the pair kernels reach their steady duration after ~80 launches"
  python tools/warmup_trend.py $(find $O/kt_trend -name "*.db" | head -1) 20; } > $O/warmup_trend.txt
rm -rf $O/kt_trend
for c in wn18rr-rotate fb15k237-complex fb15k237-transe fb15k237-distmult yago310-rotate umls-transe; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 --no-traffic 2>/dev/null | tail -1 >> $O/bench_configs.jsonl
done
# N > 1 entry path on ONE device (gloo through the host: a functional check, not a scaling number): what the driver launches
# (default partitioning table-rows, the others + config 5 in the same line), and config 5 on its own
echo "== python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (MKB_BENCH_ONE_DEVICE=1, gloo)" >> $O/one_device_world2.txt
MKB_BENCH_ONE_DEVICE=1 MKB_BENCH_EXTRAS_TIMEOUT=600 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/one_device_world2.txt
echo "== ... --config yago310-rotate --no-extras" >> $O/one_device_world2.txt
MKB_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 20 --warmup 5 --config yago310-rotate --no-extras 2>/dev/null | tail -1 >> $O/one_device_world2.txt
# what ONE rank of the row-sharded step computes (world 1: every collective degenerates to a copy), next to the plain
# single-GPU step of the same config, and its kernel trace
for extra in "" "--parallelism table-rows --force-parallelism"; do
  timeout 300 python bench.py --config yago310-rotate --no-traffic $extra 2>/dev/null | tail -1 >> $O/table_rows_one_rank.jsonl
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_tr -o run -- python $R/bench.py --config yago310-rotate --parallelism table-rows --force-parallelism --no-traffic --steps 60 --warmup 10 > /dev/null 2> $O/kt_tr.log
cd $R
python tools/prof_summary.py $(find $O/kt_tr -name "*.db" | head -1) > $O/table_rows_one_rank_kernel_stats.txt
python tools/prof_summary.py $(find $O/kt_tr -name "*.db" | head -1) timeline > $O/table_rows_one_rank_timeline.txt
rm -rf $O/kt_tr
timeout 300 python tools/general_path_speed.py > $O/general_path.txt 2>&1
for w in 1 2 4 8; do timeout 300 python tools/shard_emulate.py $w 2>&1 | grep -v amdgpu.ids >> $O/shard_emulate.txt; done
timeout 300 python tools/pipeline_speed.py 2>&1 | tail -1 > $O/pipeline_speed.txt
# the literal drop-in paths (README loop / Pipeline on the host dataset, torch.optim.Adam or mkb_amd.optim.Adam)
timeout 900 python tools/dropin_paths.py 2>/dev/null | grep -E "^#|ms/step" > $O/dropin_paths.txt
echo "# tools/readme_loop_speed.py (README loop, batches sliced on the device, 300 timed steps; the row-lazy variants are host-bound)" >> $O/dropin_paths.txt
timeout 600 python tools/readme_loop_speed.py 2>/dev/null | grep "README loop" >> $O/dropin_paths.txt
# one training step of the other configurations as kernel timelines
for c in wn18rr-rotate umls-transe fb15k237-complex fb15k237-transe yago310-rotate; do tools/timeline.sh rf_$c $c > /dev/null 2>&1; mv gpurun_out/tl_rf_$c.txt $O/timeline_$c.txt; done
# the row-sharded step through RCCL at world 1 (bench line + kernel trace with the rccl kernels)
tools/rccl_world1.sh > $O/rccl_world1.log 2>&1; cp -r gpurun_out/rccl1 $O/rccl1
rm -rf $O/ktrace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_clk   # keep the summaries, not the databases
ls -la $O

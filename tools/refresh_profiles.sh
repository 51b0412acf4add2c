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
BENCH="python $R/bench.py --no-cpu-baseline --mrr-epochs 0 --no-variants"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktrace -o run -- $BENCH --steps 60 --warmup 10 > $O/bench_under_rocprof.json 2> $O/ktrace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_$c.json 2> $O/pmc_$c.log
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $O/pmc_sq -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_sq.json 2> $O/pmc_sq.log
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_clk -o run -- $BENCH --steps 20 --warmup 5 --profile-kernel none > $O/pmc_clk.json 2> $O/pmc_clk.log
cd $R
python tools/prof_summary.py $(ls $O/ktrace/*.db | head -1) > $O/kernel_trace_stats.txt
python tools/prof_summary.py $(ls $O/ktrace/*.db | head -1) timeline > $O/kernel_timeline.txt
cp $(ls $O/ktrace/*kernel_stats.csv 2>/dev/null | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) > $O/pmc_hbm_traffic.txt
python tools/pmc_summary.py $(ls $O/pmc_sq/*.db | head -1) $(ls $O/pmc_clk/*.db | head -1) > $O/pmc_sq.txt 2>&1
timeout 900 python bench.py --breakdown > $O/bench_default.json 2> $O/bench_default.err
for c in wn18rr-rotate fb15k237-complex fb15k237-transe fb15k237-distmult yago310-rotate umls-transe; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 2>/dev/null | tail -1 >> $O/bench_configs.jsonl
done
for par in table-rows dims rows; do
  echo "== world 2 on ONE device (gloo; functional check of the N > 1 code path, not a scaling number): $par" >> $O/one_device_world2.txt
  MKB_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 3 --config yago310-rotate --parallelism $par 2>/dev/null | tail -1 >> $O/one_device_world2.txt
done
timeout 300 python tools/general_path_speed.py > $O/general_path.txt 2>&1
for w in 1 2 4 8; do timeout 300 python tools/shard_emulate.py $w 2>&1 | grep -v amdgpu.ids >> $O/shard_emulate.txt; done
timeout 300 python tools/pipeline_speed.py 2>&1 | tail -1 > $O/pipeline_speed.txt
rm -rf $O/ktrace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_clk   # keep the summaries, not the databases
ls -la $O

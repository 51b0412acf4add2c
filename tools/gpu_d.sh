#!/bin/bash
mkdir -p gpurun_out/d
{
bash tools/ab_env.sh headline pool_fwd "MKB_POOL_TILE_KS=16" "MKB_POOL_TILE_KS=4" "MKB_POOL_TILE_FSL=4" "MKB_POOL_TILE_FSL=1"
bash tools/ab_env.sh headline pool_bwd_q "MKB_POOL_PBLOCKS=8" "MKB_POOL_TPW=2"
bash tools/ab_env.sh headline adam "MKB_ADAM_SWEEP=0" "MKB_ADAM_SWEEP=16" "MKB_ADAM_SWEEP=64" "MKB_ADAM_UNROLL=4"
} 2>&1 | tee gpurun_out/d/ab.txt
# the r05 stall: does a concurrent smi poll (what the driver's gpu_busy sampler does) stall a window?
which rocm-smi amd-smi 2>&1 | tee gpurun_out/d/smi_which.txt
python bench.py --steps 20 --warmup 5 --windows 60 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants > gpurun_out/d/quiet.json 2> gpurun_out/d/quiet.err
( for i in $(seq 1 400); do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; done ) &
SMI=$!
python bench.py --steps 20 --warmup 5 --windows 60 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants > gpurun_out/d/smi_rocm.json 2> gpurun_out/d/smi_rocm.err
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
( for i in $(seq 1 400); do amd-smi metric --usage --json > /dev/null 2>&1; done ) &
SMI=$!
python bench.py --steps 20 --warmup 5 --windows 60 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants > gpurun_out/d/smi_amd.json 2> gpurun_out/d/smi_amd.err
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
python - <<'PY'
import json
for n in ("quiet","smi_rocm","smi_amd"):
    try:
        d=json.loads(open(f"gpurun_out/d/{n}.json").read().strip().splitlines()[-1])
        w=sorted(d["windows_ms_per_step"])
        print(n, "median", round(d["ms_per_step"],4), "min", w[0], "max", w[-1], "p90", w[int(len(w)*0.9)], "largest gaps", sorted(x["largest_submit_gap_ms"] for x in d["windows_detail"])[-3:])
    except Exception as e:
        print(n, "ERR", e)
PY

#!/bin/bash
# round 6, call A: new parity tests + the driver's bench protocol on fresh processes (with and without the gc freeze)
mkdir -p gpurun_out/a
python -m pytest tests/test_gpu_rank_oracle.py tests/test_gpu_general.py tests/test_gpu_defer.py -m gpu -x -q > gpurun_out/a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/a/tests.log
tail -3 gpurun_out/a/tests.log
for i in 1 2 3 4; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/a/drv_$i.json 2> gpurun_out/a/drv_$i.err
done
for i in 1 2 3; do
  MKB_BENCH_NO_GC_FREEZE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --mrr-epochs 0 --no-cpu-baseline --no-traffic --no-variants > gpurun_out/a/nofreeze_$i.json 2> gpurun_out/a/nofreeze_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/a/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['ms_per_step'],4), d['windows_ms_per_step'], 'gc', d['gc_collections_during_windows'], 'host', round(d['t_host_ms_per_step'],4))
        print('   ', [(w['largest_submit_gap_ms'], w['gc_ms'], w['device_ms']) for w in d['windows_detail']])
    except Exception as e:
        print(f, 'ERR', e)
PY

mkdir -p gpurun_out/r2a
(timeout 1500 python -m pytest tests/test_gpu_pool.py tests/test_gpu_general.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r2a/tests.txt
rm -f gpurun_out/r2a/bench.txt
for v in 1; do
  echo "BWD1=$v" >> gpurun_out/r2a/bench.txt
  MKB_POOL_BWD1=$v timeout 600 python bench.py --breakdown --no-cpu-baseline --mrr-epochs 0 >> gpurun_out/r2a/bench.txt 2>&1
done
for pbk in 8; do
  echo "BWD1=1 PBLOCKS=$pbk" >> gpurun_out/r2a/bench.txt
  MKB_POOL_PBLOCKS=$pbk timeout 600 python bench.py --breakdown --no-cpu-baseline --mrr-epochs 0 >> gpurun_out/r2a/bench.txt 2>&1
done
for c in fb15k237-transe wn18rr-rotate yago310-rotate; do for v in 1; do
  echo "cfg=$c BWD1=$v" >> gpurun_out/r2a/bench.txt
  MKB_POOL_BWD1=$v timeout 600 python bench.py --config $c --breakdown >> gpurun_out/r2a/bench.txt 2>&1
done; done
cat gpurun_out/r2a/tests.txt; grep -E "BWD1|probe|ms_per_step" gpurun_out/r2a/bench.txt | cut -c1-330

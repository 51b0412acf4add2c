mkdir -p gpurun_out/r2a
(timeout 1500 python -m pytest tests/test_gpu_pool.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r2a/tests.txt
rm -f gpurun_out/r2a/bench.txt
echo "product" >> gpurun_out/r2a/bench.txt
timeout 900 python bench.py --breakdown --mrr-epochs 0 --no-cpu-baseline --no-variants >> gpurun_out/r2a/bench.txt 2>&1
for c in fb15k237-transe wn18rr-rotate yago310-rotate; do
  echo "cfg=$c" >> gpurun_out/r2a/bench.txt
  timeout 600 python bench.py --config $c --breakdown >> gpurun_out/r2a/bench.txt 2>&1
done
cat gpurun_out/r2a/tests.txt; grep -E "product|cfg|probe|Error|error|ms_per_step" gpurun_out/r2a/bench.txt | cut -c1-330

# A/B of the three-way bf16 split GEMM (MKB_GEMM_BF16X3) on the GEMM-shaped models: parity tests with it on, then step times
MKB_GEMM_BF16X3=1 python -m pytest tests/test_gpu_pool.py tests/test_gpu_general.py -x -q -k "ComplEx or DistMult or complex or distmult or mfma or gemm" 2>&1 | tail -4
for rep in 1 2; do for v in 0 1; do for c in fb15k237-complex fb15k237-distmult; do
  echo -n "bf16x3=$v $c: "; MKB_GEMM_BF16X3=$v python bench.py --config $c --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['ms_per_step'],4), r.get('kernel','')[:50], round(r['avg_kernel_us'],1), round(r['frac'],3), 'loss', j.get('loss'))"
done; done; done

MKB_GEMM_BF16X3=1 python -m pytest tests/test_gpu_pool.py tests/test_gpu_general.py -x -q -k "ComplEx or DistMult or complex or distmult or mfma or gemm" 2>&1 | tail -2
for pipe in 1 0; do for c in fb15k237-complex fb15k237-distmult; do
  echo -n "bf16x3=1 pipe=$pipe $c: "; MKB_GEMM_PIPE=$pipe MKB_GEMM_BF16X3=1 python bench.py --config $c --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['ms_per_step'],4), round(r['avg_kernel_us'],1))"
done; done
bash tools/timeline.sh cx_bf fb15k237-complex MKB_GEMM_BF16X3=1 MKB_GEMM_PIPE=0 | head -10
bash tools/timeline.sh cx_bf1 fb15k237-complex MKB_GEMM_BF16X3=1 MKB_GEMM_PIPE=1 | head -10

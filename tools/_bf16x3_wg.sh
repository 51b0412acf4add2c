for wg in 0 400 500 800; do for c in fb15k237-complex fb15k237-distmult; do
  echo -n "bf16x3=1 min_wg=$wg $c: "; MKB_GEMM_MIN_WG=$wg MKB_GEMM_BF16X3=1 python bench.py --config $c --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['ms_per_step'],4), round(r['avg_kernel_us'],1))"
done; done

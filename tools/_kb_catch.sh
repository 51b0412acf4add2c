# catch-up kernel workgroup size / replay unroll variants (tools/kbench.py build c256:-DMKB_CATCH_THREADS=256,-DMKB_REPLAY_UNROLL=1 ...)
for rep in 1 2; do for v in "$@"; do
  for c in headline wn18rr-rotate yago310-rotate; do
    echo -n "$v $c: "; MKB_HIP_LIB=$PWD/variants/lib_$v.so python bench.py --config $c --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel sampler --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_us'],1) if j.get('roofline') else None, repr(j['loss']))"
  done
done; done

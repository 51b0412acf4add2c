export MKB_HIP_LIB=$PWD/variants/lib_w8.so
for pb in 4 8; do
  echo "== w8 MKB_POOL_PBLOCKS=$pb"; MKB_POOL_PBLOCKS=$pb python -m pytest tests/test_gpu_pool.py -q -x -k "headline_full_size_every_row or (headline_shape_pooled_equals_general and RotatE) or config2_full_size_pooled" 2>&1 | tail -3
done
unset MKB_HIP_LIB
for rep in 1 2; do
for v in base w8; do for pb in 4 8; do
  echo -n "$v pblocks=$pb headline: "; MKB_POOL_PBLOCKS=$pb MKB_HIP_LIB=$PWD/variants/lib_$v.so python bench.py --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel pool_bwd_q --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_us'],1), repr(j['loss']))"
done; done; done

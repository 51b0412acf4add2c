# A/B of library variants in the driver's protocol (--steps 20 --warmup 5): bash tools/_kb_driver.sh base fa
for rep in 1 2 3; do for v in "$@"; do
  echo -n "$v driver-protocol: "; MKB_HIP_LIB=$PWD/variants/lib_$v.so python bench.py --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4))"
done; done

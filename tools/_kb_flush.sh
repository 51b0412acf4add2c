# the dedicated flush kernel against the catch-up kernel doing the flush, in the driver's protocol (--steps 20 --warmup 5)
for rep in 1 2 3; do for v in 1 0; do
  if [ $v = 1 ]; then export MKB_ADAM_NO_FLUSH_KERNEL=1; else unset MKB_ADAM_NO_FLUSH_KERNEL; fi
  echo -n "no_flush_kernel=$v: "; python bench.py --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), repr(j['loss']))"
done; done

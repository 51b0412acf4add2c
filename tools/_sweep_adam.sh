# advance-launch sweep period (MKB_ADAM_SWEEP, adam.hip set_row_blocks) against the step time: bash tools/_sweep_adam.sh wn18rr-rotate 0 32 64 128
cfg=$1; shift
B="python bench.py --config $cfg --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel sampler --steps 400 --warmup 100"
for rep in 1 2; do for p in "$@"; do
  echo -n "$cfg sweep=$p: "; MKB_ADAM_SWEEP=$p $B 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_us'],1) if j.get('roofline') else None)"
done; done

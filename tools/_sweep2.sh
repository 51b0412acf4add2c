B="python bench.py --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel sampler --steps 600 --warmup 100"
run() { c=$1; shift; echo -n "$c $* : "; env "$@" $B --config $c 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_us'],1) if j.get('roofline') else None)"; }
for c in yago310-rotate wn18rr-rotate; do
run $c X=0
run $c MKB_ADAM_SWEEP=16
run $c MKB_ADAM_SWEEP=32
run $c MKB_ADAM_SWEEP=128
run $c MKB_ADAM_SWEEP=0
done

B="python bench.py --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --profile-kernel pool_bwd_q --steps 300 --warmup 30"
run() { c=$1; shift; echo -n "$c $* : "; env "$@" $B --config $c 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],4), round(j['roofline']['avg_kernel_us'],1))"; }
for c in wn18rr-rotate yago310-rotate; do
run $c X=0
run $c MKB_POOL_PBLOCKS=4
run $c MKB_POOL_PBLOCKS=2
run $c MKB_POOL_DENSE=0
done

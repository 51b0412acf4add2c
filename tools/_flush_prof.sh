R=$(pwd); O=$R/gpurun_out/flushprof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export MKB_ADAM_NO_FLUSH_KERNEL=1; else unset MKB_ADAM_NO_FLUSH_KERNEL; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/k$v -o run -- python $R/bench.py --no-traffic --no-cpu-baseline --mrr-epochs 0 --no-variants --steps 20 --warmup 5 --profile-kernel none > /dev/null 2> $O/log$v.txt
  python $R/tools/prof_summary.py $(find $O/k$v -name "*.db" | head -1) | grep -E "^kernel|adam_rows" 
done

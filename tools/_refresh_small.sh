# the part of tools/refresh_profiles.sh that the GEMM change touches: the other configurations' bench lines and the timelines
# of the two GEMM-shaped models
R=$(pwd); O=$R/gpurun_out/refresh_small; rm -rf $O; mkdir -p $O
for c in wn18rr-rotate fb15k237-complex fb15k237-transe fb15k237-distmult yago310-rotate umls-transe; do
  timeout 300 python bench.py --config $c --no-cpu-baseline --mrr-epochs 0 --no-traffic 2>/dev/null | tail -1 >> $O/bench_configs.jsonl
done
for c in fb15k237-complex fb15k237-distmult; do tools/timeline.sh rf_$c $c > /dev/null 2>&1; mv gpurun_out/tl_rf_$c.txt $O/timeline_$c.txt; done
tools/timeline.sh rf_cx_fp32 fb15k237-complex MKB_GEMM_BF16X3=0 > /dev/null 2>&1; mv gpurun_out/tl_rf_cx_fp32.txt $O/timeline_fb15k237-complex_fp32_kernel.txt
ls -la $O

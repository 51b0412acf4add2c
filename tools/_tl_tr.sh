R=$(pwd); O=$R/gpurun_out/tl_tr; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o run -- python $R/bench.py --config yago310-rotate --parallelism table-rows --force-parallelism --no-traffic --steps 60 --warmup 10 > /dev/null 2> $O/log.txt
cd $R; python tools/prof_summary.py $(find $O -name "*.db" | head -1) timeline; rm -rf $O

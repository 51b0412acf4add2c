#!/bin/bash
# round 5: timelines of the small / TransE configurations + the kernel stats of the unchanged README loop with torch.optim.Adam
R=$(pwd)
tools/timeline.sh r5_umls umls-transe > /dev/null 2>&1
tools/timeline.sh r5_transe fb15k237-transe > /dev/null 2>&1
tools/timeline.sh r5_headline headline > /dev/null 2>&1
O=$R/gpurun_out/r5_readme; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o run -- python $R/tools/readme_loop_speed.py > $O/out.txt 2> $O/log.txt
cd $R
db=$(find $O -name "*.db" | head -1)
python tools/prof_summary.py $db | head -45 > $R/gpurun_out/r5_readme_kernels.txt
cat $O/out.txt >> $R/gpurun_out/r5_readme_kernels.txt
rm -rf $O
cat gpurun_out/tl_r5_umls.txt gpurun_out/tl_r5_transe.txt gpurun_out/tl_r5_headline.txt gpurun_out/r5_readme_kernels.txt

#!/bin/bash
# rocprofv3 kernel statistics of the README loop with ONE optimizer variant (MKB_README_ONLY, default the row-lazy one)
R=$(pwd); O=$R/gpurun_out/readme_prof; rm -rf $O; mkdir -p $O
export MKB_README_ONLY="${MKB_README_ONLY:-mkb_amd.optim.Adam(lazy_rows)}"
python tools/readme_loop_speed.py 2>/dev/null | tail -1 > $O/speed.txt; cat $O/speed.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o run -- python $R/tools/readme_loop_speed.py > /dev/null 2> $O/kt.log
cd $R
python tools/prof_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
rm -rf $O/kt
head -50 $O/kernel_stats.txt | cut -c1-170

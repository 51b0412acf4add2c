#!/bin/bash
# On the GPU box: the row-sharded step (mkb_amd/table_rows.py) with every collective issued through RCCL at world size 1
# (bench.py --rccl-world1: `nccl` process group + MKB_ROWS_FORCE_COLLECTIVES=1), next to the same path with the collectives
# short-circuited and to the plain single-GPU step, then its rocprofv3 kernel trace (RCCL kernels / copies in the timeline).
R=$(pwd); O=$R/gpurun_out/rccl1; rm -rf $O; mkdir -p $O
B="python $R/bench.py --config yago310-rotate --no-traffic --steps 200 --warmup 20"
for extra in "" "--parallelism table-rows --force-parallelism" "--rccl-world1"; do
  timeout 300 $B $extra > $O/run.log 2>&1; grep -a "^{\"metric\"" $O/run.log | tail -1 >> $O/bench.jsonl; grep -a -i "error\|Traceback" $O/run.log | head -5
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -o run -- python $R/bench.py --config yago310-rotate --rccl-world1 --no-traffic --steps 60 --warmup 10 > /dev/null 2> $O/kt.log
cd $R
db=$(find $O/kt -name "*.db" | head -1)
python tools/prof_summary.py $db > $O/kernel_stats.txt 2>&1
python tools/prof_summary.py $db timeline > $O/timeline.txt 2>&1
python - <<P >> $O/timeline.txt 2>&1
import sqlite3
con = sqlite3.connect("$db")
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if 'memory_cop' in t.lower()]
print("# memory-copy records:", {t: con.execute(f"select count(*) from {t}").fetchone()[0] for t in mc})
P
rm -rf $O/kt
python - <<P
import json
for l in open("$O/bench.jsonl"):
    j = json.loads(l); print(round(j["ms_per_step"], 4), j["config"]["parallelism"][:40], j.get("table_rows"))
P
head -40 $O/timeline.txt; head -24 $O/kernel_stats.txt

"""Per-kernel mean duration by launch order from a rocprofv3 kernel trace of bench.py: which launch of the step gets faster over the
first hundreds of steps of a process (the windows_ms trend of the bench line)?
    python tools/warmup_trend.py <results.db> [launches per bucket]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = con.execute("select name, start, end from kernels order by start").fetchall()
by = {}
for name, s, e in rows:
    k = re.sub(r"\(.*$", "", name).replace("void ", "")[:60]
    by.setdefault(k, []).append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 3 * per:
        continue
    buckets = [sum(v[i: i + per]) / len(v[i: i + per]) for i in range(0, len(v) - per + 1, per)]
    print(f"{k:62s} n={len(v):5d} " + " ".join(f"{b:6.1f}" for b in buckets[:16]))

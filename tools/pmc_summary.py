"""Per-kernel mean of a rocprofv3 --pmc counter (rocpd SQLite):  python tools/pmc_summary.py db [db ...]
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced streams (MI355X_MICROARCH.md, HBM section): the 'x2' column applies that correction."""
import re
import sqlite3
import sys


def short(n):
    return re.sub(r"\(.*$", "", n).replace("void ", "")[:70]


for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    agg = {}
    for name, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        a = agg.setdefault((short(name), cname), [0, 0.0])
        a[0] += 1
        a[1] += val
    print(f"# {path}")
    print(f"{'kernel':72s} {'counter':12s} {'calls':>6s} {'mean_MB':>10s} {'x2_MB':>10s}")
    for (k, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if c.endswith("_SIZE"):
            mb = tot / n * 1024 / 1e6
            print(f"{k:72s} {c:12s} {n:6d} {mb:10.3f} {2 * mb if c == 'FETCH_SIZE' else float('nan'):10.3f}")
        else:  # plain event counters (SQ_*): mean count per dispatch
            print(f"{k:72s} {c:22s} {n:6d} {tot / n:16.0f}")

"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a per-kernel table:
    python tools/prof_summary.py gpurun_out/prof/run_results.db > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:112s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / tot:6.2f}")


def timeline(path, step_kernel="adam_rows_catchup_kernel", which=-3):
    """One training step as a timeline: start offset, duration and the idle gap before every kernel."""
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if step_kernel in r[0]]
    if len(marks) < 4:  # small tables step densely: the step then opens with the sampler's draw
        marks = [i for i, r in enumerate(rows) if "pool_draw_kernel" in r[0]]
    lo, hi = marks[which], marks[which + 1]
    t0, prev_end, busy = rows[lo][1], rows[lo][1], 0
    print(f"{'kernel':70s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}")
    for name, s, e in rows[lo:hi]:
        print(f"{short(name)[:70]:70s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
        busy += e - s
        prev_end = e
    print(f"step {(rows[hi][1] - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {hi - lo} launches")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "timeline":
        timeline(sys.argv[1])
    else:
        main(sys.argv[1])

#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (kernel trace) into a small per-kernel CSV summary.
usage: summarize_rocprof.py <results.db> <out.csv> [top_n]"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in rows[:top]:
        short = name if len(name) < 160 else name[:157] + "..."
        w.writerow([short, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])
print(f"wrote {out} ({min(top, len(rows))} of {len(rows)} kernels)")

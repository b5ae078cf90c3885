#!/usr/bin/env python3
"""Summarise rocprofv3 results (.db, rocpd schema) into small CSV files.

  summarize_rocprof.py trace <results.db> <out.csv> [top_n]   per-kernel totals (kernel trace)
  summarize_rocprof.py pmc   <results.db> <out.csv> [filter]  per (kernel, grid) average counter
                                                             values (PMC pass)
The .db files are tens of MB; only these summaries are kept under profiles/.
"""
import csv
import sqlite3
import sys


def trace(db, out, top=25):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent"])  # top_kernels durations are in us
        for name, calls, total, avg, pct in rows[:top]:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, f"{total / 1e3:.3f}", f"{avg / 1e3:.3f}", f"{pct:.3f}"])
    print(f"wrote {out} ({min(top, len(rows))} of {len(rows)} kernels)")


def pmc(db, out, flt="dann"):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, grid_size, workgroup_size, lds_block_size, vgpr_count, sgpr_count, counter_name, "
         "count(*), avg(value), avg(duration) from counters_collection where kernel_name like ? "
         "group by kernel_name, grid_size, counter_name order by avg(duration) desc")
    rows = list(cur.execute(q, (f"%{flt}%",)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "workgroup", "lds_bytes", "vgprs", "sgprs", "counter", "dispatches",
                    "avg_value", "avg_duration_us"])
        for r in rows:
            name = r[0] if len(r[0]) < 140 else r[0][:137] + "..."
            w.writerow([name, r[1], r[2], r[3], r[4], r[5], r[6], r[7], f"{r[8]:.3f}", f"{r[9] / 1e3:.3f}"])
    print(f"wrote {out} ({len(rows)} rows)")


if __name__ == "__main__":
    mode, db, out = sys.argv[1], sys.argv[2], sys.argv[3]
    if mode == "trace":
        trace(db, out, int(sys.argv[4]) if len(sys.argv) > 4 else 25)
    else:
        pmc(db, out, sys.argv[4] if len(sys.argv) > 4 else "dann")

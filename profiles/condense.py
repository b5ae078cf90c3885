#!/usr/bin/env python3
"""Condense one profiles/run_profiles.sh pass (gpurun_out/<tag>_*) into the committed summaries:
  profiles/<name>_kernel_trace.csv, <name>_bench*.json, <name>_pmc_beam_search.csv, <name>_pmc_summary.json
and refresh profiles/pmc_latest.json (read by bench.py for roofline.traffic).
Usage: python profiles/condense.py <tag> <name>
"""
import csv, glob, json, os, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = f"{R}/gpurun_out", f"{R}/profiles"

def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])

bench = last_json(f"{G}/{tag}_bench.json")
under = last_json(f"{G}/{tag}_bench_under_rocprof.json")
nq = 100000
for tok in bench["config"]["workload"].split():
    if tok.replace(",", "").isdigit() and "queries/step" in bench["config"]["workload"].split(tok, 1)[1][:14]:
        nq = int(tok.replace(",", ""))
grid = nq * 64
shutil.copy(f"{G}/{tag}_bench.json", f"{P}/{name}_bench.json")
shutil.copy(f"{G}/{tag}_bench_under_rocprof.json", f"{P}/{name}_bench_under_rocprof.json")
shutil.copy(f"{G}/{tag}_kernel_trace.csv", f"{P}/{name}_kernel_trace.csv")

trace = None
for row in csv.DictReader(open(f"{G}/{tag}_kernel_trace.csv")):
    if "beam_search_kernel<0, 0, false, 1, 128" in row["kernel"]:
        trace = row
rows, vals, durs = [], {}, []
for path in sorted(glob.glob(f"{G}/{tag}_pmc_*.csv")):
    for row in csv.DictReader(open(path)):
        if "beam_search_kernel" in row["kernel"] and int(row["grid_size"]) == grid:
            rows.append(row)
            vals[row["counter"]] = float(row["avg_value"])
            durs.append(float(row["avg_duration_us"]))
with open(f"{P}/{name}_pmc_beam_search.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
hbm = (fetch_kb * 2 + write_kb) * 1024
summary = {
    "source": f"profiles/run_profiles.sh {tag}: rocprofv3 --pmc (one counter group per pass, --kernel-trace only) over "
              "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --L <chosen>`; rows of the timed "
              f"launch (grid {nq} workgroups x 64)",
    "workload": {"nq": nq, "L": bench["config"]["L"], "beam_width": bench["config"]["beam_width"], "n": 1000000,
                 "dim": 128},
    "kernel": "beam_search_kernel<F32, L2, QS=1, DIM=128, MODE=plain>",
    "FETCH_SIZE_kb_per_launch": fetch_kb,
    "WRITE_SIZE_kb_per_launch": write_kb,
    "fetch_correction": "x2 on gfx950 for 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE "
                        "uncalibrated",
    "hbm_bytes_per_launch_corrected": hbm,
    "avg_duration_us_under_pmc": sum(durs) / len(durs),
    "TCC_HIT_sum": vals.get("TCC_HIT_sum"),
    "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
    "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
    "SQ": {k: v for k, v in vals.items() if k.startswith("SQ_")},
    "GRBM_GUI_ACTIVE": vals.get("GRBM_GUI_ACTIVE"),
    "bench_line_avg_kernel_ms_plain_run": bench["roofline"]["avg_kernel_ms"],
    "bench_line_avg_kernel_ms_under_rocprof_trace": under["roofline"]["avg_kernel_ms"],
    "kernel_trace_avg_ms_same_command": float(trace["avg_ms"]),
    "kernel_trace_calls": int(trace["calls"]),
}
json.dump(summary, open(f"{P}/{name}_pmc_summary.json", "w"), indent=1)
json.dump(summary, open(f"{P}/pmc_latest.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_launch_corrected", "l2_hit_rate", "avg_duration_us_under_pmc",
                                          "bench_line_avg_kernel_ms_plain_run",
                                          "bench_line_avg_kernel_ms_under_rocprof_trace",
                                          "kernel_trace_avg_ms_same_command", "kernel_trace_calls")}, indent=1))

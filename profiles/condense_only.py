#!/usr/bin/env python3
"""Condense one profiles/run_only.sh pass (gpurun_out/<tag>_<workload>_*) into profiles/<name>_<workload>_summary.json
(+ the kernel-trace CSV and the PMC rows of the dominant kernel).  For the large index it also refreshes
profiles/pmc_large_latest.json, which bench.py reads for roofline_large.traffic.
Usage: python profiles/condense_only.py <tag> <workload> <name>"""
import csv, glob, json, os, shutil, sys

tag, wl, name = sys.argv[1], sys.argv[2], sys.argv[3]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = f"{R}/gpurun_out", f"{R}/profiles"
flt = "expand_beam" if wl == "gather" else "search_kernel"  # beam_search_kernel and pair_search_kernel
if wl in ("large_u8", "large_sq8", "u8", "sq8"):  # (their index is built under the profiler too: the build's insert searches
    flt = "pair_search_kernel"                    #  are beam_search_kernel launches, as long as or longer than the timed searches)


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


plain = list(last_json(f"{G}/{tag}_{wl}.json").values())[0]
under = list(last_json(f"{G}/{tag}_{wl}_under_rocprof.json").values())[0]


def timed_leg(obj):
    """the 10 M integer legs (large_u8 / large_sq8, run with --L 64) nest their timed leg: {.., "L64": {..}}"""
    for key, val in obj.items():
        if key.startswith("L") and key[1:].isdigit() and isinstance(val, dict) and "avg_kernel_ms" in val:
            return {**{k: v for k, v in obj.items() if not isinstance(v, dict)}, **val}
    return obj


plain, under = timed_leg(plain), timed_leg(under)
shutil.copy(f"{G}/{tag}_{wl}_kernel_trace.csv", f"{P}/{name}_{wl}_kernel_trace.csv")
trace = None
for row in csv.DictReader(open(f"{G}/{tag}_{wl}_kernel_trace.csv")):
    # the timed launches are the longest ones of that kernel family (the index build launches many short ones)
    # (a single longer dispatch of the family -- the first, uncalibrated launch -- is not the timed workload)
    key = lambda r: (int(r["calls"]) >= 5, float(r["avg_ms"]))
    if flt in row["kernel"] and (trace is None or key(row) > key(trace)):
        trace = row
rows, vals, durs = [], {}, []
for path in sorted(glob.glob(f"{G}/{tag}_{wl}_pmc_*.csv")):
    best = {}
    for row in csv.DictReader(open(path)):  # the timed launch = the longest dispatch group of that kernel
        if flt in row["kernel"]:
            c = row["counter"]
            k2 = lambda r: (int(r["dispatches"]) >= 5, float(r["avg_duration_us"]))
            if c not in best or k2(row) > k2(best[c]):
                best[c] = row
    for c, row in best.items():
        rows.append(row)
        vals[c] = float(row["avg_value"])
        durs.append(float(row["avg_duration_us"]))
with open(f"{P}/{name}_{wl}_pmc.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
hbm = (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024
alg = plain.get("algorithmic_bytes_per_launch") or plain["search_kernel"]["algorithmic_bytes_per_launch"]  # (pq: the search kernel alone)
summary = {
    "source": f"profiles/run_only.sh {tag} {wl}: plain run, then rocprofv3 --kernel-trace --stats, then one --pmc pass per "
              "counter group over the same command (`python bench.py --only " + wl + " ...`)",
    "plain_run": plain,
    "under_kernel_trace": {k: under[k] for k in ("avg_kernel_ms",) if k in under} or
                          {"avg_kernel_ms": under.get("search_kernel", {}).get("avg_kernel_ms")},
    "kernel_trace_dominant_kernel": trace,
    "FETCH_SIZE_kb_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": vals["WRITE_SIZE"],
    "fetch_correction": "x2 on gfx950 for 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated",
    "hbm_bytes_per_launch_corrected": hbm,
    "traffic_over_algorithmic": hbm / alg,
    "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
    "l2_hit_rate": (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])) if "TCC_HIT_sum" in vals else None,
    "avg_duration_us_under_pmc": sum(durs) / len(durs),
}
json.dump(summary, open(f"{P}/{name}_{wl}_summary.json", "w"), indent=1)
if wl in ("large",):
    spec = sys.argv[4] if len(sys.argv) > 4 else "10000000:128:sift_like:1:2560:32:28:100"
    json.dump({"source": f"profiles/{name}_{wl}_summary.json", "workload": spec, "L": plain["L"],
               "nq": int(plain["workload"].split(" queries/launch")[0].split()[-1]),
               "hbm_bytes_per_launch_corrected": hbm}, open(f"{P}/pmc_large_latest.json", "w"), indent=1)
if wl == "pq":  # bench.py reads this for pq.search_kernel_traffic (rocprofv3 cannot run inside bench.py)
    sk = plain["search_kernel"]
    json.dump({"source": f"profiles/{name}_{wl}_summary.json (profiles/run_only.sh {tag} pq: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum "
                         "TCC_MISS_sum in separate --pmc passes over `python bench.py --only pq --L <L>`; rows of the timed launch)",
               "workload": {"nq": 100000, "L": plain["L"], "n": 1000000, "dim": 128, "chunks": plain["chunks"]},
               "kernel": sk.get("kernel"), "kernel_family": sk.get("kernel_family"),
               "packed_neighbor_codes": plain.get("packed_neighbor_codes") is not None,
               "FETCH_SIZE_kb_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": vals["WRITE_SIZE"],
               "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM section), as for the headline kernel",
               "fabric_bytes_per_launch_corrected": hbm, "algorithmic_bytes_per_launch": alg,
               "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
               "l2_hit_rate": summary["l2_hit_rate"], "avg_duration_us_under_pmc": summary["avg_duration_us_under_pmc"],
               "kernel_trace_avg_ms": float(trace["avg_ms"]) if trace else None,
               "queries_per_cu": sk.get("queries_per_cu"), "traffic_over_algorithmic": hbm / alg},
              open(f"{P}/pmc_pq_latest.json", "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("hbm_bytes_per_launch_corrected", "traffic_over_algorithmic", "l2_hit_rate",
                                          "avg_duration_us_under_pmc")}, indent=1), trace)

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05i}
t0=$(date +%s)
timeout 1200 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3))
oc=d["other_configs"]
for k in ("single_query_L64_latency","concurrent_1024_qps_at_L","sustained_1024_in_flight_qps_at_L"):
    print(k, oc.get(k) if not isinstance(oc.get(k), dict) else {kk: oc[k][kk] for kk in list(oc[k])[:6]})
for k in ("sq8","u8","pq"):
    print(k, json.dumps(oc.get(k))[:900])
print("hbm_side", json.dumps(d["roofline"].get("hbm_side"))[:1500])
for k in ("roofline_large","roofline_large_d768","roofline_large_d768_f16"):
    r=d.get(k,{})
    print(k, {kk: r.get(kk) for kk in ("L","recall_at_10","qps","frac","bound","avg_kernel_ms","kernel_family","error")}, r.get("oracle_sample"))
print("mfma", d.get("mfma"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","kind")})
PY

#!/bin/bash
# round 5, sixth lease: bucket-of-two visited table + advisor fixes -- whole suite, PQ / u8 / sq8 legs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05f}
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/${T}_pytest_all.log 2>&1
grep -E "passed|failed|error|Fatal|Memory access|^FAILED|Error" gpurun_out/${T}_pytest_all.log | head -8
for w in pq u8 sq8; do timeout 400 python bench.py --only $w > gpurun_out/${T}_$w.json 2> gpurun_out/${T}_$w.err; done
python - <<PY
import json
for w in ("pq","u8","sq8"):
    try:
        d=json.loads(open("gpurun_out/${T}_%s.json"%w).read().strip().splitlines()[-1])[w]
        sk=d.get("search_kernel", d)
        print(w, "L",d["L"],"family",sk.get("kernel_family"),"kernel ms",round(sk["avg_kernel_ms"],3),"qps",round(sk.get("qps_search_only", sk.get("qps",0))),"frac",round(sk.get("frac_of_hbm_peak",0),3),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"),d["oracle_sample"].get("distances_cmps_hops_identical"))
    except Exception as e: print(w, "error", e)
PY

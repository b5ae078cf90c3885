#!/bin/bash
# round 5, fifth lease: whole suite after the PQ kernel + touch_row fix; PQ leg; counter names for DRAM-side evidence
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=r05e
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/${T}_pytest_all.log 2>&1
grep -E "passed|failed|error|Fatal|Memory access" gpurun_out/${T}_pytest_all.log | head -5
timeout 400 python bench.py --only pq > gpurun_out/${T}_pq_packed.json 2> gpurun_out/${T}_pq_packed.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_pq_packed.json").read().strip().splitlines()[-1])["pq"]
sk=d["search_kernel"]
print("packed", "L",d["L"],"recall",d["recall_at_10_vs_exact_f32"],"family",sk.get("kernel_family"),"kernel ms",round(sk["avg_kernel_ms"],3),"qps_search_only",round(sk["qps_search_only"]),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"),d["oracle_sample"].get("distances_cmps_hops_identical"))
PY
(rocprofv3 --list-avail 2>/dev/null || rocprofv3 -L 2>/dev/null) > gpurun_out/${T}_counters_avail.txt 2>&1
grep -i -c "name" gpurun_out/${T}_counters_avail.txt
grep -i -o -E "\b[A-Z_0-9]*(DRAM|MALL|HBM|EA0?_RD)[A-Za-z_0-9\[\]]*" gpurun_out/${T}_counters_avail.txt | sort -u | head -60

#!/bin/bash
# round 5, eighth lease: load limit of the 16-bit visited tables (6 vs 7 eighths) on the u8 / PQ kernels; whole suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05j}
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/${T}_pytest_all.log 2>&1
grep -E "passed|failed|error|Fatal|Memory access|^FAILED|Error|assert" gpurun_out/${T}_pytest_all.log | head -12
for e in 6 7 5; do
for L in 26 64; do DANN_HT16_OPEN_EIGHTHS=$e DANN_VERBOSE=1 timeout 400 python bench.py --only u8 --L $L > gpurun_out/${T}_u8_L${L}_e$e.json 2> gpurun_out/${T}_u8_L${L}_e$e.err; done
DANN_HT16_OPEN_EIGHTHS=$e timeout 400 python bench.py --only pq --L 96 > gpurun_out/${T}_pq_e$e.json 2> /dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_*_e*.json")):
    try:
        d=list(json.loads(open(f).read().strip().splitlines()[-1]).values())[0]
        sk=d.get("search_kernel", d)
        print(f.split("/")[-1], "L",d["L"],"family",sk.get("kernel_family"),"kernel ms",round(sk["avg_kernel_ms"],3),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"),d["oracle_sample"].get("distances_cmps_hops_identical"))
    except Exception as e: print(f, "error", e)
PY
grep -h "visited cap" gpurun_out/${T}_u8_L64_e*.err | sort | uniq -c | head

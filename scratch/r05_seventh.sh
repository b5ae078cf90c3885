#!/bin/bash
# round 5, seventh lease: pair kernel with two queue entries / two adjacency ids per lane -- suite, u8 at L = 26 / 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05g}
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -x > gpurun_out/${T}_pytest_all.log 2>&1
grep -E "passed|failed|error|Fatal|Memory access|^FAILED|Error|assert" gpurun_out/${T}_pytest_all.log | head -12
for L in 26 64; do timeout 400 python bench.py --only u8 --L $L > gpurun_out/${T}_u8_L$L.json 2> gpurun_out/${T}_u8_L$L.err; done
DANN_TUNE_OFF=16 timeout 400 python bench.py --only u8 --L 64 > gpurun_out/${T}_u8_L64_onewave.json 2> /dev/null
timeout 400 python bench.py --only sq8 --L 64 > gpurun_out/${T}_sq8_L64.json 2> /dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_*8_L*.json")):
    try:
        d=list(json.loads(open(f).read().strip().splitlines()[-1]).values())[0]
        print(f.split("/")[-1], "L",d["L"],"family",d.get("kernel_family"),"kernel ms",round(d["avg_kernel_ms"],3),"qps",round(d["qps"]),"frac",round(d["frac_of_hbm_peak"],3),"cmps",round(d["mean_cmps"]),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"),d["oracle_sample"].get("distances_cmps_hops_identical"))
    except Exception as e: print(f, "error", e)
PY

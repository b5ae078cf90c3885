#!/bin/bash
# u8 pair kernel: table margin 10 % / 5 % x load limit 6 / 7 eighths, L = 26 / 48 / 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05m}
for cfg in "6 0" "7 0" "7 2" "6 2"; do set -- $cfg
for L in 26 48 64; do DANN_HT16_OPEN_EIGHTHS=$1 DANN_TUNE_ON=$2 DANN_VERBOSE=1 timeout 300 python bench.py --only u8 --L $L > gpurun_out/${T}_u8_L${L}_e$1_m$2.json 2> gpurun_out/${T}_u8_L${L}_e$1_m$2.err; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_u8_*.json")):
    try:
        d=list(json.loads(open(f).read().strip().splitlines()[-1]).values())[0]
        print(f.split("/")[-1], "L",d["L"],"family",d.get("kernel_family"),"kernel ms",round(d["avg_kernel_ms"],3),"frac",round(d["frac_of_hbm_peak"],3),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"))
    except Exception as e: print(f, "error", e)
PY
grep -h "two queries per wavefront" gpurun_out/${T}_u8_L64_e*.err | sort | uniq -c

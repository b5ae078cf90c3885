#!/bin/bash
# u8 pair kernel at L = 64: explicit visited tables smaller than the calibrated one (more wavefronts per CU, more spills)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05k}
for w in 0 640 768 896 1024 1152; do
DANN_VERBOSE=1 timeout 300 python bench.py --only u8 --L 64 --visited-bits $w > gpurun_out/${T}_u8_L64_w$w.json 2> gpurun_out/${T}_u8_L64_w$w.err
done
timeout 300 python bench.py --only u8 --L 26 --visited-bits 512 > gpurun_out/${T}_u8_L26_w512.json 2> /dev/null
timeout 300 python bench.py --only u8 --L 26 --visited-bits 640 > gpurun_out/${T}_u8_L26_w640.json 2> /dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_u8_*.json")):
    try:
        d=list(json.loads(open(f).read().strip().splitlines()[-1]).values())[0]
        print(f.split("/")[-1], "L",d["L"],"family",d.get("kernel_family"),"kernel ms",round(d["avg_kernel_ms"],3),"frac",round(d["frac_of_hbm_peak"],3),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"))
    except Exception as e: print(f, "error", e)
PY
grep -h "two queries per wavefront" gpurun_out/${T}_u8_L64_w*.err | sort | uniq -c | head
timeout 200 python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_u8_L64_w0.json").read().strip().splitlines()[-1])["u8"]
print("hops", d["mean_hops"], "cmps", d["mean_cmps"])
PY

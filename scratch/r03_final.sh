#!/bin/bash
# Round-3 evidence on one box: parity suite + smoke, default bench line, headline trace + PMC, build trace + MFMA counters
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r03g}
cd $R; mkdir -p gpurun_out
bash scratch/final_run.sh $T   # full: tests, smoke, bench, headline profiles, u8 / sq8 profiles
cd /tmp && export TMPDIR=/tmp
A="1000000 768 64 56 128 16384"
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $R/gpurun_out/${T}_build768.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $R/gpurun_out/${T}_build768_kernel_trace.csv 12 > /dev/null 2>&1
python $R/profiles/condense_build.py $R/gpurun_out/${T}_build768.log $R/gpurun_out/${T}_build768_kernel_trace.csv $R/gpurun_out/${T}_build768_summary.json > /dev/null 2>&1
for C in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm/p_$N -o p -- python $R/scratch/build_phases.py $A > /dev/null 2>&1
    python $R/profiles/summarize_rocprof.py pmc /tmp/pm/p_$N/p_results.db $R/gpurun_out/${T}_build768_pmc_$N.csv gram_tiles > /dev/null 2>&1
done
timeout 300 python $R/scratch/build_phases.py $A --f16 > $R/gpurun_out/${T}_build768_f16.log 2>&1
tail -1 $R/gpurun_out/${T}_build768.log; tail -1 $R/gpurun_out/${T}_build768_f16.log
python - <<PY
import json
o=json.load(open("$R/gpurun_out/${T}_build768_summary.json"))
print({k:round(v["total_ms"]) for k,v in o["kernels"].items()}); print(o["rates"].get("gram_tiles_kernel")); print(o["model"]["build_seconds"], o["model"]["mfma"]["mfma_share_of_all_prune_pair_distances"])
PY
ls $R/gpurun_out | grep $T | wc -l

#!/bin/bash
# kernel trace of the 1 M x 768 f32 build (per-kernel totals) + insert-search launch parameters
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04j}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="1000000 768 64 56 128 16384"
rm -rf /tmp/pm && DANN_DEBUG=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $O/build768.log 2> $O/build768.err
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $O/build768_kernel_trace.csv 14 > /dev/null 2>&1
python $R/profiles/condense_build.py $O/build768.log $O/build768_kernel_trace.csv $O/build768_summary.json > /dev/null 2>&1
python - <<PY
import json
o=json.load(open("$O/build768_summary.json"))
print({k:(round(v["total_ms"]), v["calls"]) for k,v in o["kernels"].items()}); print(o["rates"]); print(o["model"]["build_seconds"])
PY
grep "visited cap" $O/build768.err | sort | uniq -c | sort -rn | head -5

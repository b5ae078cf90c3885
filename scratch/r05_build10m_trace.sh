#!/bin/bash
# kernel trace of the 10 M x 128 build (inside `bench.py --only large --L 56`): where do its 10 s go?
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05r}
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
D=/tmp/prof_$T; rm -rf $D; mkdir -p $D
timeout 900 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $R/bench.py --only large --L 56 > $R/gpurun_out/${T}_large_under_rocprof.json 2> $D/err.log
python $R/profiles/summarize_rocprof.py trace $D/trace/t_results.db $R/gpurun_out/${T}_build10m_kernel_trace.csv 25
cut -c1-200 $R/gpurun_out/${T}_build10m_kernel_trace.csv | head -30
python - <<PY
import json
d=json.loads([l for l in open("$R/gpurun_out/${T}_large_under_rocprof.json").read().splitlines() if l.startswith("{")][-1])["roofline_large"]
print(d["workload"][:300]); print(d.get("build"))
PY

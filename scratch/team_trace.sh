#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r03z}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tt/t -o t -- python $R/scratch/team_trace.py > $R/gpurun_out/${T}_team_trace.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/tt/t/t_results.db $R/gpurun_out/${T}_team_kernel_trace.csv 12 > /dev/null 2>&1
grep "team kernel" $R/gpurun_out/${T}_team_trace.log; grep beam_search $R/gpurun_out/${T}_team_kernel_trace.csv | cut -c1-220

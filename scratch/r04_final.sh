#!/bin/bash
# Round-4 evidence on one box: parity suite + smoke, default bench line, headline trace + PMC, u8 / sq8 / pq / large passes
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04z}
cd $R; mkdir -p gpurun_out
bash scratch/final_run.sh $T
for w in pq large; do timeout 400 bash profiles/run_only.sh $T $w > gpurun_out/${T}_only_$w.log 2>&1; done
ls gpurun_out | grep $T | wc -l

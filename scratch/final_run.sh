#!/bin/bash
# Round-end evidence on one box: [parity suite + smoke,] default bench line, headline trace + PMC, [u8 / sq8 profiles]
# usage: scratch/final_run.sh <tag> [bench-only]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r02k}
if [ -z "${2:-}" ]; then
timeout 300 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
fi
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -2 gpurun_out/${T}_bench.err
cut -c1-300 gpurun_out/${T}_bench.json
SKIP_PLAIN=1 PMC_SHORT=1 timeout 500 bash profiles/run_profiles.sh $T > gpurun_out/${T}_profiles.log 2>&1; tail -3 gpurun_out/${T}_profiles.log
if [ -z "${2:-}" ]; then
for w in u8 sq8; do timeout 200 bash profiles/run_only.sh $T $w > gpurun_out/${T}_only_$w.log 2>&1; done
fi
ls gpurun_out | grep $T | wc -l

#!/bin/bash
# Round-6 evidence on one box: parity suite + smoke, default bench line, headline trace + PMC, u8 / sq8 / pq / large passes.
# usage: scratch/r06_final.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r06z}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tee gpurun_out/${T}_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/${T}_pytest.log
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -2 gpurun_out/${T}_bench.err
cut -c1-300 gpurun_out/${T}_bench.json
SKIP_PLAIN=1 PMC_SHORT=1 timeout 600 bash profiles/run_profiles.sh $T > gpurun_out/${T}_profiles.log 2>&1; tail -3 gpurun_out/${T}_profiles.log
for w in u8 sq8 pq; do timeout 400 bash profiles/run_only.sh $T $w > gpurun_out/${T}_only_$w.log 2>&1; done
timeout 400 bash profiles/run_only.sh ${T}L64 u8 --L 64 > gpurun_out/${T}_only_u8_L64.log 2>&1
timeout 900 bash profiles/run_only.sh $T large --L 56 > gpurun_out/${T}_only_large.log 2>&1
ls $R/gpurun_out | grep $T | wc -l

#!/bin/bash
# Round-end evidence on one box: parity suite + smoke, default bench line, headline trace + PMC, u8 / sq8 profiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T=${1:-r02k}
timeout 300 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
/usr/bin/time -f "bench wall %e s" timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -2 gpurun_out/${T}_bench.err
cut -c1-300 gpurun_out/${T}_bench.json
SKIP_PLAIN=1 PMC_SHORT=1 timeout 500 bash profiles/run_profiles.sh $T > gpurun_out/${T}_profiles.log 2>&1; tail -3 gpurun_out/${T}_profiles.log
for w in u8 sq8; do timeout 200 bash profiles/run_only.sh $T $w > gpurun_out/${T}_only_$w.log 2>&1; done
ls gpurun_out | grep $T | wc -l

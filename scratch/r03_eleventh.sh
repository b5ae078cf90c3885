#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03m
mkdir -p $OUT
cd $R
timeout 300 python scratch/team_repro.py > $OUT/repro.log 2>&1
grep -v "^/opt" $OUT/repro.log | tail -8
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
timeout 400 python scratch/team_lab.py 1000000 > $OUT/team_lab.log 2>&1
grep -v "^{" $OUT/team_lab.log | grep -E "team_spec|one_wave_nq1_|one_wave_nq1024" | tail -9
timeout 300 python scratch/latency_lab.py --prof --tunes 0 --points 1:64:300,1:26:300 > $OUT/phases.log 2>&1
grep -v "^/opt" $OUT/phases.log | tail -5

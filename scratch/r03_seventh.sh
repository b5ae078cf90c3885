#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
timeout 400 python scratch/team_lab.py 1000000 > $OUT/team_lab.log 2>&1
grep -v "^{" $OUT/team_lab.log | tail -45

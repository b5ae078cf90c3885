#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
cd $R
timeout 180 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=120 -p no:cacheprovider -k "team" > $OUT/pytest_team.log 2>&1
rc=$?; echo "team test rc=$rc"; tail -3 $OUT/pytest_team.log
if [ $rc -ne 0 ]; then tail -40 $OUT/pytest_team.log; exit 1; fi
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_server.py tests/test_gpu_edges.py -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
timeout 400 python scratch/team_lab.py 1000000 > $OUT/team_lab.log 2>&1
grep -v "^{" $OUT/team_lab.log | grep -E "nq1_|nq16_|nq1024_" | tail -15
timeout 300 python scratch/latency_lab.py --prof --tunes 0 --points 1:64:300,1:26:300 > $OUT/phases.log 2>&1
grep -v "^/opt" $OUT/phases.log | tail -5

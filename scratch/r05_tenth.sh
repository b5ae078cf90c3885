#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05y}
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${T}_pytest_all.log 2>&1
grep -E "passed|failed|error|Fatal|Memory access|^FAILED|Error|assert" gpurun_out/${T}_pytest_all.log | head -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scratch/r05_bench.sh $T 2>&1 | head -12

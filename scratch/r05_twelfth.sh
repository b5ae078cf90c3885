#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05t}
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed|Fatal|Memory access"
bash scratch/r05_build10m_trace.sh $T 2>&1 | head -14

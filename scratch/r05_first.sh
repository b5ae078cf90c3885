#!/bin/bash
# round 5, first lease: the whole GPU suite with the kernel-family assertions; default-threshold pair test first in its own process
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -15 > gpurun_out/r05a_pytest.log
echo "pytest wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r05a_pytest.log
cat gpurun_out/r05a_pytest.log

#!/bin/bash
# round 6, fourth call (re-entry): the whole GPU suite at the head + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tee -a $O/pytest.log

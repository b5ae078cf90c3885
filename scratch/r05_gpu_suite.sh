#!/bin/bash
# the whole GPU suite + smoke, results to gpurun_out/<tag>/pytest.log
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r05suite}; O=$R/gpurun_out/$T; mkdir -p $O
shift
timeout 1500 python -m pytest ${@:-tests} -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tee -a $O/pytest.log

#!/bin/bash
# round 6, first call: the GPU suite on the advisor fixes + the default bench line as the round's baseline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tee -a $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json

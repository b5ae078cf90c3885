#!/bin/bash
# round 6, tenth call: whole GPU suite (filtered tie order, lane pipeline of the host-pointer call, wide PQ tables) + the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -5; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06j/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_hbm_side'))
print(json.dumps(d['other_configs']['host_pointer_search_batch'])[:900])
PY

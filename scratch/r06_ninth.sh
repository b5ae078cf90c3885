#!/bin/bash
# round 6, ninth call: host-pointer pipeline with the helper thread / pinned caller buffers -- test, then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_server.py -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_hbm_side'))
oc=d['other_configs']
print(json.dumps(oc['host_pointer_search_batch'])[:900])
print(json.dumps(oc['pq'])[:300])
PY

#!/bin/bash
# round 6, fifteenth call: temporary page-locking of reused pageable buffers -- server tests, host-pipeline sweep, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06p; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_server.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python scratch/r06_hostpipe.py 2>&1 | grep -v amdgpu.ids | head -8 | tee $O/hostpipe.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'])
print(json.dumps(d['other_configs']['host_pointer_search_batch'])[:1400])
PY

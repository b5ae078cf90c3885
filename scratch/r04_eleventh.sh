#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_server.py tests/test_gpu_range.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
timeout 900 python bench.py --large none > $O/bench.json 2> $O/bench.err
python - <<PY
import json
o=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k,v in o["other_configs"]["concurrent_callers"].items(): print(k, v)
PY

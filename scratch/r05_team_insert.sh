#!/bin/bash
# team-of-four kernel for the insert searches of small batches: parity tests, then the 1 M x 128 build time
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r05t}; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > $O/pytest.txt 2>&1; grep -E "passed|failed|Error" $O/pytest.txt | tail -5
for off in 0 4; do
DANN_TUNE_OFF=$off timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_$off.json 2> $O/bench_$off.err
python - <<PY
import json
b=json.loads([l for l in open("$O/bench_$off.json").read().splitlines() if l.startswith("{")][-1])
print("tune_off $off: 1M x 128 build", b["config"]["build_seconds"], "recall", b["config"]["recall_at_10"], "L", b["config"]["L"], "value", round(b["value"]))
PY
done

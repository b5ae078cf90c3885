#!/bin/bash
# prune sweep with the next candidates' rows requested ahead: build parity, 10 M x 128 / 1 M x 128 / u8 build times, trace
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05w}
timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py tests/test_gpu_quant.py -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed"
timeout 600 python bench.py --only large --L 56 > gpurun_out/${T}_large.json 2> gpurun_out/${T}_large.err
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${T}_1m.json 2> gpurun_out/${T}_1m.err
timeout 300 python bench.py --only u8 > gpurun_out/${T}_u8.json 2> gpurun_out/${T}_u8.err; grep -h "build" gpurun_out/${T}_u8.err | tail -2
python - <<PY
import json,re
d=json.loads([l for l in open("gpurun_out/${T}_large.json").read().splitlines() if l.startswith("{")][-1])["roofline_large"]
b=json.loads([l for l in open("gpurun_out/${T}_1m.json").read().splitlines() if l.startswith("{")][-1])
print("10M build", re.search(r"built on the GPU in ([0-9.]+) s", d["workload"]).group(1), "s | 1M build", b["config"]["build_seconds"], "s recall", b["config"]["recall_at_10"])
PY
bash scratch/r05_build10m_trace.sh $T 2>&1 | sed -n 2,8p | cut -c1-150

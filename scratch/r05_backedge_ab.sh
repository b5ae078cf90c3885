#!/bin/bash
# back-edge phase of small-row builds: one kernel sized by the batch's longest list vs scan + short / long worklists,
# by the pool size up to which the single-kernel form is kept (DANN_BACKEDGE_SINGLE_POOL; 8192 = round 4's behaviour)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05s}
timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py -m gpu -q --timeout 300 2>&1 | grep -E "passed|failed"
for p in 8192 512 256 128 64; do
  DANN_BACKEDGE_SINGLE_POOL=$p timeout 600 python bench.py --only large --L 56 > gpurun_out/${T}_large_p$p.json 2> gpurun_out/${T}_large_p$p.err
  DANN_BACKEDGE_SINGLE_POOL=$p timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${T}_1m_p$p.json 2> gpurun_out/${T}_1m_p$p.err
  python - <<PY
import json,re
d=json.loads([l for l in open("gpurun_out/${T}_large_p$p.json").read().splitlines() if l.startswith("{")][-1])["roofline_large"]
b=json.loads([l for l in open("gpurun_out/${T}_1m_p$p.json").read().splitlines() if l.startswith("{")][-1])
print("single_pool $p: 10M build", re.search(r"built on the GPU in ([0-9.]+) s", d["workload"]).group(1), "s  search", round(d["avg_kernel_ms"],2), "ms | 1M build", b["config"]["build_seconds"], "s recall", b["config"]["recall_at_10"], "L", b["config"]["L"])
PY
done

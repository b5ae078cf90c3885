#!/bin/bash
# after vectorising the prune's consume loop: build parity tests, 10 M x 128 build with a kernel trace, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r05v}; O=$R/gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py -m gpu -q --timeout 300 > $O/pytest.txt 2>&1; grep -E "passed|failed" $O/pytest.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- python $R/bench.py --only large --L 56 > $O/large.json 2> $O/large.err
python $R/profiles/summarize_rocprof.py trace /tmp/kt/k_results.db $O/large_kernel_stats.csv 14 > /dev/null 2>&1
ls /tmp/kt | head
head -12 $O/large_kernel_stats.csv 2>/dev/null
cd $R
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json,re
d=json.loads([l for l in open("$O/large.json").read().splitlines() if l.startswith("{")][-1])["roofline_large"]
print("10M build", re.search(r"built on the GPU in ([0-9.]+) s", d["workload"]).group(1), "s search", round(d["avg_kernel_ms"],2))
b=json.loads([l for l in open("$O/bench.json").read().splitlines() if l.startswith("{")][-1])
print("1M build", b["config"]["build_seconds"], "recall", b["config"]["recall_at_10"], "value", b["value"])
for k,v in b.items():
    if isinstance(v,dict) and "workload" in v and "build" in str(v.get("workload")): print(k, str(v["workload"])[:200])
PY

#!/bin/bash
# round 5, second lease: PQ lookup-table kernel -- parity suite, then the PQ leg packed / plain / old kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=r05b
timeout 600 python -m pytest tests/test_gpu_pqlut.py tests/test_gpu_quant.py -m gpu -q --timeout 300 -x 2>&1 | tail -25 > gpurun_out/${T}_pytest_pq.log
cat gpurun_out/${T}_pytest_pq.log | tail -8
DANN_VERBOSE=1 timeout 400 python bench.py --only pq > gpurun_out/${T}_pq_packed.json 2> gpurun_out/${T}_pq_packed.err; tail -3 gpurun_out/${T}_pq_packed.err
timeout 400 python bench.py --only pq --no-pq-pack > gpurun_out/${T}_pq_plain.json 2> gpurun_out/${T}_pq_plain.err
DANN_TUNE_OFF=32 timeout 400 python bench.py --only pq --no-pq-pack > gpurun_out/${T}_pq_old.json 2> gpurun_out/${T}_pq_old.err
for f in packed plain old; do python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_pq_$f.json").read().strip().splitlines()[-1])["pq"]
sk=d["search_kernel"]
print("$f", "L",d["L"],"recall",d["recall_at_10_vs_exact_f32"],"family",sk.get("kernel_family"),"kernel ms",round(sk["avg_kernel_ms"],3),"qps_search_only",round(sk["qps_search_only"]),"oracle",d["oracle_sample"])
PY
done
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -5 > gpurun_out/${T}_pytest_all.log; cat gpurun_out/${T}_pytest_all.log

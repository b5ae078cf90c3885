#!/bin/bash
# round 5, fourth lease: PQ kernel with the hand-scheduled table lookups -- parity, timing, counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=r05d
timeout 600 python -m pytest tests/test_gpu_pqlut.py tests/test_gpu_quant.py tests/test_gpu_pair.py -m gpu -q --timeout 300 -x 2>&1 | tail -8 > gpurun_out/${T}_pytest_pq.log
cat gpurun_out/${T}_pytest_pq.log | tail -4
timeout 400 python bench.py --only pq > gpurun_out/${T}_pq_packed.json 2> gpurun_out/${T}_pq_packed.err
timeout 400 python bench.py --only pq --no-pq-pack > gpurun_out/${T}_pq_plain.json 2> gpurun_out/${T}_pq_plain.err
for f in packed plain; do python - <<PY
import json
d=json.loads(open("gpurun_out/${T}_pq_$f.json").read().strip().splitlines()[-1])["pq"]
sk=d["search_kernel"]
print("$f", "L",d["L"],"recall",d["recall_at_10_vs_exact_f32"],"family",sk.get("kernel_family"),"kernel ms",round(sk["avg_kernel_ms"],3),"qps_search_only",round(sk["qps_search_only"]),"oracle",d["oracle_sample"].get("ids_identical_to_gpu"),d["oracle_sample"].get("distances_cmps_hops_identical"), "hops", d["mean_hops"], "cmps", d["mean_cmps"])
PY
done
bash scratch/r05_pq_pmc.sh ${T}_pmc

#!/bin/bash
# the PQ lookup-table kernel at 10 M points (160 MB of codes, 1.3 GB adjacency, 7 GB packed layout): packed / plain / old kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05p}
t0=$(date +%s)
timeout 900 python bench.py --only pq --n 10000000 --dist sift_like:1:2560 --max-batch 65536 > gpurun_out/${T}_pq10m_packed.json 2> gpurun_out/${T}_pq10m_packed.err; echo "rc=$? wall $(( $(date +%s) - t0 )) s"; tail -2 gpurun_out/${T}_pq10m_packed.err
L=$(python -c "import json;print(json.loads(open('gpurun_out/${T}_pq10m_packed.json').read().strip().splitlines()[-1])['pq']['L'])")
timeout 900 python bench.py --only pq --n 10000000 --dist sift_like:1:2560 --max-batch 65536 --L $L --no-pq-pack > gpurun_out/${T}_pq10m_plain.json 2> /dev/null
DANN_TUNE_OFF=32 timeout 900 python bench.py --only pq --n 10000000 --dist sift_like:1:2560 --max-batch 65536 --L $L --no-pq-pack > gpurun_out/${T}_pq10m_old.json 2> /dev/null
python - <<PY
import json
for f in ("packed","plain","old"):
    try:
        d=json.loads(open("gpurun_out/${T}_pq10m_%s.json"%f).read().strip().splitlines()[-1])["pq"]
        sk=d["search_kernel"]
        print(f, "L",d["L"],"recall",d["recall_at_10_vs_exact_f32"],"family",sk.get("kernel_family"),"kernel ms",round(sk["avg_kernel_ms"],3),"qps_search_only",round(sk["qps_search_only"]),"cmps",round(d["mean_cmps"]),"hops",round(d["mean_hops"],1),"oracle",d["oracle_sample"], "pack", d.get("packed_neighbor_codes"))
    except Exception as e: print(f,"error",e)
PY

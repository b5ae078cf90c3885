#!/bin/bash
# round 4, third GPU call: full GPU suite, pair A/B after the sink change, the default bench line, the PQ profile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -25 > $O/pytest_gpu.log
for t in 16 0; do DANN_TUNE_OFF=$t timeout 200 python bench.py --only u8 2>/dev/null | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('tune_off=$t u8', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak')}, v.get('oracle_sample'))"; done > $O/ab_pair.log 2>&1
t0=$(date +%s)
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s" > $O/bench_wall.txt
timeout 600 bash profiles/run_only.sh r04e pq > $O/only_pq.log 2>&1

#!/bin/bash
# headline QPS at recall@10 >= 0.95 as a function of the graph's build parameters (1 M x 128 f32)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04l; mkdir -p $O
for cfg in "32 28 100" "32 28 200" "32 30 100" "40 36 100" "48 42 100" "64 56 128" "24 21 100" "32 28 64"; do set -- $cfg
  timeout 300 python bench.py --no-cpu-baseline --no-extras --large none --max-degree $1 --pruned-degree $2 --l-build $3 2>/dev/null | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); c=o['config']
print('R=$1 pruned=$2 l_build=$3', 'QPS', round(o['value']/1e6,2), 'L', c['L'], 'recall', c['recall_at_10'], 'cmps', round(c['mean_cmps'],1), 'hops', round(c['mean_hops'],1), 'build_s', c['build_seconds'], 'frac', round(o['roofline']['frac'],3))"
done > $O/graph_sweep.txt 2>&1

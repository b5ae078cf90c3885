#!/bin/bash
# config 5's per-GPU shape (100 M x 768 f16 on one GPU) with the hierarchical mixture
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04g; mkdir -p $O
DANN_DEBUG=1 timeout 2400 python bench.py --only build768 --build-spec 100000000:768:64:56:128:f16 > $O/build_100m.json 2> $O/build_100m.err
tail -1 $O/build_100m.json | python -c "
import sys, json
o=json.loads(sys.stdin.read())['build_large']
print({k:o[k] for k in ('build_seconds','points_per_second','batches','data_seconds')}, o['search'], o.get('oracle_replay'), o['prune'])" > $O/summary.txt 2>&1
grep "visited cap\|build768" $O/build_100m.err | tail -6 >> $O/summary.txt

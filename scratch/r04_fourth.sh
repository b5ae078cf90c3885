#!/bin/bash
# config-5 generator at 10 M x 768 f16 with the blob count of the 100 M run: i.i.d. centres vs hierarchical centres
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04f; mkdir -p $O
for c in hier iid; do
  DANN_DEBUG=1 timeout 900 python bench.py --only build768 --build-spec 10000000:768:64:56:128:f16 --build-blobs 25601 --build-centres $c > $O/build_$c.json 2> $O/build_$c.err
  tail -1 $O/build_$c.json | python -c "
import sys, json
o=json.loads(sys.stdin.read())['build_large']
print('$c', {k:o[k] for k in ('build_seconds','points_per_second')}, {k:v for k,v in o['search'].items() if k in ('recall_at_10','L','qps','mean_cmps','frac_of_hbm_peak','recall_by_L')}, o.get('oracle_replay'))"
  grep "visited cap" $O/build_$c.err | tail -2
done > $O/summary.txt 2>&1

#!/bin/bash
# round 6, eighth call: 1 M x 768 -- PQ-32 / 48 / 64 + Rerank beside the f16 rows at full precision (same graph)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06h; mkdir -p $O
for C in 48 32 64; do
  timeout 600 python bench.py --only pq768 --pq-chunks $C > $O/pq768_$C.json 2> $O/pq768_$C.err
  tail -c 1800 $O/pq768_$C.json; echo; tail -3 $O/pq768_$C.err
done

#!/bin/bash
# round 6: config 5's per-GPU shape again -- 100 M x 768 f16 on one GPU, default tie order (DANN_TIE_RUST), f16 rows on the f16 matrix core
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06_100m; mkdir -p $O
timeout 1500 python bench.py --only build768 --build-spec 100000000:768:64:56:128:f16 > $O/build_100m.json 2> $O/build_100m.err
tail -5 $O/build_100m.err; cut -c1-1500 $O/build_100m.json

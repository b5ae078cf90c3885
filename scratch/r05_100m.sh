#!/bin/bash
# config 5's per-GPU shape with the round-5 build kernels (pipelined prune, striped statistics): 100 M x 768 f16 on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05m}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
DANN_DEBUG=1 timeout 560 python bench.py --only build768 --build-spec 100000000:768:64:56:128:f16 > $O/build_100m.json 2> $O/build_100m.err
echo rc=$?; tail -c 600 $O/build_100m.json; grep "build768" $O/build_100m.err | tail -5

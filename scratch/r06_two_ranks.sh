#!/bin/bash
# two ranks on ONE GPU (DANN_BENCH_ONE_DEVICE=1): the driver's multi-GPU launch line, as far as one device goes -- RCCL
# communicator, sharded build, per-rank QPS, the shared set's byte identity
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06s; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 DANN_BENCH_ONE_DEVICE=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
echo rc=$?; tail -5 $O/bench_2ranks.err | cut -c1-300; tail -c 1500 $O/bench_2ranks.json

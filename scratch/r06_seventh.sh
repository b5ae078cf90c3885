#!/bin/bash
# round 6, seventh call: PQ lookup-table kernel for 17 .. 64 chunks + the trainer's path counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06g; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_pqlut.py tests/test_gpu_quant.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -15 $O/pytest.log

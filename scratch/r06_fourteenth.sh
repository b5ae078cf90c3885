#!/bin/bash
# round 6, fourteenth call: the server / host-pipeline tests at the head (poisoned-server quiesce, concurrent host batches)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_server.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log

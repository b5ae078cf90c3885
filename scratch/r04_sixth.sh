#!/bin/bash
# range / filtered / paged / server tests after the shared-lock change; latency of the pair kernel at small batches
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_range.py tests/test_gpu_filtered.py tests/test_gpu_server.py tests/test_gpu_edges.py tests/test_gpu_sharding.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -15 > $O/pytest.log
timeout 600 python scratch/pair_latency.py > $O/pair_latency.txt 2>&1

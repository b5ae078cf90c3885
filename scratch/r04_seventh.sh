#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_range.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest.log
PAIR_LAT_NQ=4096,6144,8192,12288,16384,24576,32768,65536 timeout 600 python scratch/pair_latency.py > $O/pair_latency.txt 2>&1

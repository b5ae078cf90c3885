#!/bin/bash
# build parity after the four-targets-per-wave back-edge scan, then the 768-d build trace again
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04k; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py tests/test_gpu_quant.py tests/test_gpu_formats.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_build.log
bash scratch/r04_build_prof.sh r04k > $O/build_prof.txt 2>&1

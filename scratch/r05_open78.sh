#!/bin/bash
# 16-bit visited table open to 7/8 instead of 6/8 of its slots: does the insert search of 3 KB rows gain a residency step?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r05o}; O=$R/gpurun_out/$T; mkdir -p $O
for e in 6 7; do
  for spec in 1000000:768:64:56:128:f32 1000000:768:64:56:128:f16; do
    DANN_HT16_OPEN_EIGHTHS=$e DANN_VERBOSE=1 timeout 300 python bench.py --only build768 --build-spec $spec > $O/b_${e}_${spec##*:}.json 2> $O/b_${e}_${spec##*:}.err
    echo "open $e/8 $spec: $(grep -E 'build [0-9.]+s' $O/b_${e}_${spec##*:}.err | tail -1) | $(grep -E 'L=128 W=1: visited cap' $O/b_${e}_${spec##*:}.err | tail -1)"
  done
done
DANN_HT16_OPEN_EIGHTHS=7 timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py tests/test_gpu_visited16.py -m gpu -q --timeout 300 > $O/pytest7.txt 2>&1; grep -E "passed|failed" $O/pytest7.txt
timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py -m gpu -q --timeout 300 > $O/pytest6.txt 2>&1; grep -E "passed|failed" $O/pytest6.txt

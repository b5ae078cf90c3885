#!/bin/bash
# round 6, third call: overflow table of the 16-bit visited tables -- pair / PQ tests, then the 10 M integer legs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pair.py tests/test_gpu_pqlut.py tests/test_gpu_visited16.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for K in u8 sq8; do
  timeout 900 python bench.py --only large_$K > $O/large_$K.json 2> $O/large_$K.err
  tail -3 $O/large_$K.err
done
timeout 600 python bench.py --only u8 --L 64 > $O/u8_L64.json 2> $O/u8_L64.err
timeout 600 python bench.py --only u8 --L 26 > $O/u8_L26.json 2> $O/u8_L26.err

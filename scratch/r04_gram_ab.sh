#!/bin/bash
# NOTE: the DANN_BUILD_ONE_STREAM / DANN_BUILD_ITEM_ORDER / DANN_GRAM_ONE_KERNEL switches this A/B used were removed from the library after the
# measurement (results: profiles/r04q_*); check out commit 1e8b133 to repeat it.
# gram_tiles variants on the 1 M x 768 build: wall clock + HIP-event time of the tile launches, and the build tests
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04s}; O=$R/gpurun_out/$T; mkdir -p $O
A=${2:-"1000000 768 64 56 128 16384"}
cd /tmp
for i in 1 2; do
  for V in "X=0" "DANN_GRAM_ONE_KERNEL=1"; do
    echo "$V:" $(env $V timeout 200 python $R/scratch/build_phases.py $A 2>/dev/null | grep -o "build [0-9.]*s\|gram_tiles.*" | tr '\n' ' ')
  done
done 2>&1 | tee $O/ab.txt
(cd $R && timeout 900 python -m pytest tests/test_gpu_build.py -x -q 2>&1 | tail -3) | tee $O/pytest_build.log
(cd $R && DANN_GRAM_ONE_KERNEL=1 timeout 900 python -m pytest tests/test_gpu_build.py -x -q 2>&1 | tail -3) | tee $O/pytest_build_w2.log

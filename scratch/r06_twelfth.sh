#!/bin/bash
# round 6, twelfth call: 10 M x 128 u8 / SQ-8 at L = 64 under rocprofv3 -- kernel trace, then FETCH / WRITE / TCC passes (each pass bounded)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for K in large_u8 large_sq8; do
  timeout 900 bash profiles/run_only.sh r06 $K --L 64 > gpurun_out/r06_only_$K.log 2>&1
  ls gpurun_out | grep "r06_$K" | tr '\n' ' '; echo
done

#!/bin/bash
# round 6, eleventh call: host-pointer pipeline -- lanes x chunk size, pageable and page-locked caller buffers
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06k; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python scratch/r06_hostpipe.py > $O/hostpipe.txt 2>&1; cat $O/hostpipe.txt | grep -v amdgpu.ids

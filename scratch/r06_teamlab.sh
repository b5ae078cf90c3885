#!/bin/sh
# latency-regime kernel times (HIP events around the small launches): one query, 16 and 1024 queries per launch
export PYTHONPATH=$PWD
timeout 600 python scratch/latency_lab.py --tunes 0 --points 1:64:300,1:26:300,16:26:200,16:64:200,256:26:100,1024:26:100,1024:64:50 2>&1 | grep -v amdgpu

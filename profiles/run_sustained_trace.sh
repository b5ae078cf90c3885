#!/bin/bash
# kernel trace of the default bench extras (includes the 1024-in-flight runs: the PERSIST instantiation of the search kernel)
# Usage: profiles/run_sustained_trace.sh <tag>  -> gpurun_out/<tag>_sustained_kernel_trace.csv
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sus_$TAG && mkdir -p /tmp/prof_sus_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sus_$TAG/trace -o t -- python $R/bench.py --no-cpu-baseline --large none \
    > $OUT/${TAG}_sustained_bench.json 2> /tmp/prof_sus_$TAG/trace.err
python $R/profiles/summarize_rocprof.py trace /tmp/prof_sus_$TAG/trace/t_results.db $OUT/${TAG}_sustained_kernel_trace.csv 40

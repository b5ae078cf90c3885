#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05v}
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
D=/tmp/prof_$T; rm -rf $D; mkdir -p $D
timeout 900 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $R/bench.py --only large768 --L 22 > $R/gpurun_out/${T}_large768_under_rocprof.json 2> $D/err.log
python $R/profiles/summarize_rocprof.py trace $D/trace/t_results.db $R/gpurun_out/${T}_build768_kernel_trace.csv 25
cut -c1-170 $R/gpurun_out/${T}_build768_kernel_trace.csv | head -22

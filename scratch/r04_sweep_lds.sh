#!/bin/bash
# compact LDS layout of the batched sweep: build wall clock, kernel trace, build tests
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04zu}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="1000000 768 64 56 128 16384"
for i in 1 2; do timeout 200 python $R/scratch/build_phases.py $A 2>/dev/null | grep -o "build [0-9.]*s" | head -1; done
rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $O/b.log 2> $O/b.err
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $O/kernel_trace.csv 8 > /dev/null 2>&1
cut -c1-120 $O/kernel_trace.csv | head -7
cd $R && timeout 300 python -m pytest tests/test_gpu_build.py -q --timeout 100 2>&1 | grep -E "passed|failed"

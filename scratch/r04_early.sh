#!/bin/bash
# the geometric phase of the 1 M x 768 build alone (first 330 k points: every batch below max_batch): wall clock vs kernel time
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04v}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 330000 100000 30000; do
  timeout 200 python $R/scratch/build_phases.py $n 768 64 56 128 16384 2>/dev/null | grep -o "n=.*" | head -1
done | tee $O/early.txt
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py 330000 768 64 56 128 16384 > $O/b.log 2> $O/b.err
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $O/early_kernel_trace.csv 24 > /dev/null 2>&1
grep -o "n=.*" $O/b.log | head -1
cut -c1-150 $O/early_kernel_trace.csv

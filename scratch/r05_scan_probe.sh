#!/bin/bash
# EXPERIMENT (not committed code): where backedge_scan_kernel spends its time at 10 M points
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05scan}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 16 48; do
  rm -rf /tmp/kts
  DANN_TUNE_ON=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -o k -- python $R/bench.py --only large --L 56 > $O/large_$v.json 2> $O/large_$v.err
  python $R/profiles/summarize_rocprof.py trace /tmp/kts/k_results.db $O/trace_$v.csv 8 > /dev/null 2>&1
  echo "probe $v:"; grep -E "scan|backedge_kernel" $O/trace_$v.csv | cut -c1-160
done

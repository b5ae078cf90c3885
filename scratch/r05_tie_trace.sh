#!/bin/bash
# Kernel trace of a 200 k x 128 u8 build (integer distances: tied pools) under DANN_TIE_POSITION and DANN_TIE_RUST:
# which kernels pay for the one-lane walk of tied pools, and how much.  usage: scratch/r05_tie_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05t}
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for o in position rust; do
  D=/tmp/tt_$o; rm -rf $D
  timeout 120 rocprofv3 --kernel-trace --stats -d $D -o t -- python $R/scratch/tie_order_cost.py 200000 u8 $o > $R/gpurun_out/${T}_u8_$o.json 2> $R/gpurun_out/${T}_u8_$o.err
  python $R/profiles/summarize_rocprof.py trace $D/t_results.db $R/gpurun_out/${T}_u8_${o}_kernel_trace.csv 10
  cat $R/gpurun_out/${T}_u8_$o.json; echo
done
head -8 $R/gpurun_out/${T}_u8_position_kernel_trace.csv | cut -c1-150; head -8 $R/gpurun_out/${T}_u8_rust_kernel_trace.csv | cut -c1-150

#!/bin/bash
# first GPU pass of round 3: parity suite, the build A/B (row kernel vs three-kernel MFMA pool prune), concurrency lab
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for mode in default nopool rowonly; do
  case $mode in
    default) ENVV="";;
    nopool) ENVV="DANN_POOL_GRAM=0";;
    rowonly) ENVV="";;
  esac
  EXTRA=""; [ $mode = rowonly ] && EXTRA="--rowonly"
  env $ENVV timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 $EXTRA > $OUT/build768_$mode.log 2>&1
  tail -1 $OUT/build768_$mode.log
done
env timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 --f16 > $OUT/build768_f16_default.log 2>&1; tail -1 $OUT/build768_f16_default.log
env DANN_POOL_GRAM=0 timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 --f16 > $OUT/build768_f16_nopool.log 2>&1; tail -1 $OUT/build768_f16_nopool.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_trace.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $OUT/build768_kernel_trace.csv 10 > /dev/null 2>&1
python $R/profiles/condense_build.py $OUT/build768_trace.log $OUT/build768_kernel_trace.csv $OUT/build768_summary.json > $OUT/condense.log 2>&1
cd $R
timeout 600 python scratch/server_lab.py 1000000 26 > $OUT/server_lab.log 2>&1
tail -3 $OUT/server_lab.log

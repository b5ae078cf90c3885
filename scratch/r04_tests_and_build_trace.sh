#!/bin/bash
# the whole GPU suite + smoke, then the kernel trace of the 1 M x 768 f32 and f16 builds
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04zy}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -4 | tee $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
cd /tmp && export TMPDIR=/tmp
for V in f32 f16; do
  A="1000000 768 64 56 128 16384"; [ $V = f16 ] && A="$A --f16"
  rm -rf /tmp/pm && DANN_DEBUG=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $O/build768_$V.log 2> $O/build768_$V.err
  python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $O/build768_${V}_kernel_trace.csv 16 > /dev/null 2>&1
  python $R/profiles/condense_build.py $O/build768_$V.log $O/build768_${V}_kernel_trace.csv $O/build768_${V}_summary.json > /dev/null 2>&1
  grep -o "n=.*\|gram_tiles.*" $O/build768_$V.log | head -2
done

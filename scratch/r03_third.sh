#!/bin/bash
# third GPU pass: build-path tests, 1 M x 768 A/B of the sweep prefetch, the 10 M x 768 f32 build leg
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_quant.py tests/test_gpu_formats.py -m gpu -q --maxfail=5 --timeout=300 -p no:cacheprovider > $OUT/pytest_build.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_build.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_default.log 2>&1; tail -1 $OUT/build768_default.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_default2.log 2>&1; tail -1 $OUT/build768_default2.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_trace.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $OUT/build768_kernel_trace.csv 10 > /dev/null 2>&1
python $R/profiles/condense_build.py $OUT/build768_trace.log $OUT/build768_kernel_trace.csv $OUT/build768_summary.json > $OUT/condense.log 2>&1
grep -A3 "pool_sweep\|gram_tiles" $OUT/build768_summary.json | head -20
cd $R
timeout 900 python bench.py --only build768 --build-spec 10000000:768:64:56:128:f32 > $OUT/build_10m_f32.json 2> $OUT/build_10m_f32.err
echo "build10m rc=$?"; tail -4 $OUT/build_10m_f32.err; cut -c1-1800 $OUT/build_10m_f32.json

#!/bin/bash
# sixth GPU pass: batched exact evaluations + lists up to 256 entries on the Gram path; then the 100 M x 768 f16 build
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py -m gpu -q --maxfail=5 --timeout=300 -p no:cacheprovider > $OUT/pytest_build.log 2>&1
PRC=$?
echo "pytest rc=$PRC"; tail -3 $OUT/pytest_build.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_default.log 2>&1; tail -1 $OUT/build768_default.log
DANN_BACKEDGE_GRAM_ROWS=96 timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_rows96.log 2>&1; tail -1 $OUT/build768_rows96.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 --f16 > $OUT/build768_f16.log 2>&1; tail -1 $OUT/build768_f16.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_trace.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $OUT/build768_kernel_trace.csv 10 > /dev/null 2>&1
python $R/profiles/condense_build.py $OUT/build768_trace.log $OUT/build768_kernel_trace.csv $OUT/build768_summary.json > $OUT/condense.log 2>&1
python - <<'PY'
import json,os
o=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03f/build768_summary.json"))
print({k:round(v["total_ms"]) for k,v in o["kernels"].items()}); print(o["rates"].get("gram_tiles_kernel")); print(o["model"]["mfma"])
PY
cd $R
if [ $PRC -ne 0 ]; then echo "build tests failed: skipping the 100 M run"; exit 0; fi
timeout 2400 python bench.py --only build768 --build-spec 100000000:768:64:56:128:f16 > $OUT/build_100m_f16.json 2> $OUT/build_100m_f16.err
echo "build100m rc=$?"; tail -3 $OUT/build_100m_f16.err; cut -c1-3000 $OUT/build_100m_f16.json

#!/bin/bash
# fourth GPU pass: build-path tests (batched sweep, three-kernel back-edges), 1 M x 768 A/B, the 10 M x 768 f32 build leg
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_sharding.py -m gpu -q --maxfail=5 --timeout=300 -p no:cacheprovider > $OUT/pytest_build.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_build.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_default.log 2>&1; tail -1 $OUT/build768_default.log
DANN_SWEEP_ONE_BY_ONE=1 timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_onebyone.log 2>&1; tail -1 $OUT/build768_onebyone.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 --f16 > $OUT/build768_f16.log 2>&1; tail -1 $OUT/build768_f16.log
timeout 300 python scratch/build_phases.py 1000000 128 32 28 100 16384 > $OUT/build128_default.log 2>&1; tail -1 $OUT/build128_default.log
timeout 300 python scratch/build_phases.py 1000000 128 32 28 100 16384 --mfma > $OUT/build128_mfma.log 2>&1; tail -1 $OUT/build128_mfma.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_trace.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $OUT/build768_kernel_trace.csv 10 > /dev/null 2>&1
python $R/profiles/condense_build.py $OUT/build768_trace.log $OUT/build768_kernel_trace.csv $OUT/build768_summary.json > $OUT/condense.log 2>&1
python - <<'PY'
import json,os
o=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03d/build768_summary.json"))
print({k:round(v["total_ms"]) for k,v in o["kernels"].items()}); print(o["rates"]); print(o["model"]["mfma"])
PY
cd $R
timeout 900 python bench.py --only build768 --build-spec 10000000:768:64:56:128:f32 > $OUT/build_10m_f32.json 2> $OUT/build_10m_f32.err
echo "build10m rc=$?"; tail -4 $OUT/build_10m_f32.err; cut -c1-2500 $OUT/build_10m_f32.json

#!/bin/bash
# fifth GPU pass: centred Gram -- build-path tests, 1 M x 768 A/B + trace, 10 M x 768 f32 leg, 10 M x 128 row-prefetch A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_build.py -m gpu -q --maxfail=5 --timeout=300 -p no:cacheprovider > $OUT/pytest_build.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_build.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_default.log 2>&1; tail -1 $OUT/build768_default.log
DANN_GRAM_CENTRE=0 timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_uncentred.log 2>&1; tail -1 $OUT/build768_uncentred.log
timeout 300 python scratch/build_phases.py 1000000 768 64 56 128 16384 --f16 > $OUT/build768_f16.log 2>&1; tail -1 $OUT/build768_f16.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py 1000000 768 64 56 128 16384 > $OUT/build768_trace.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $OUT/build768_kernel_trace.csv 10 > /dev/null 2>&1
python $R/profiles/condense_build.py $OUT/build768_trace.log $OUT/build768_kernel_trace.csv $OUT/build768_summary.json > $OUT/condense.log 2>&1
python - <<'PY'
import json,os
o=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03e/build768_summary.json"))
print({k:round(v["total_ms"]) for k,v in o["kernels"].items()}); print(o["rates"]); print(o["model"]["mfma"])
PY
cd $R
timeout 900 python bench.py --only build768 --build-spec 10000000:768:64:56:128:f32 > $OUT/build_10m_f32.json 2> $OUT/build_10m_f32.err
echo "build10m rc=$?"; tail -2 $OUT/build_10m_f32.err; python -c "
import json;o=json.load(open('$OUT/build_10m_f32.json'))['build_large'];print(o['build_seconds'],o['prune'],o['search']['recall_at_10'],o['search']['qps'],o['oracle_replay'])"
timeout 600 python bench.py --only large > $OUT/large_default.json 2> $OUT/large_default.err; python -c "
import json;o=json.load(open('$OUT/large_default.json'))['roofline_large'];print('large default',o['L'],o['qps'],o['frac'],o['avg_kernel_ms'])"
DANN_TUNE_ON=1 timeout 600 python bench.py --only large > $OUT/large_prefetch.json 2> $OUT/large_prefetch.err; python -c "
import json;o=json.load(open('$OUT/large_prefetch.json'))['roofline_large'];print('large prefetch',o['L'],o['qps'],o['frac'],o['avg_kernel_ms'])"

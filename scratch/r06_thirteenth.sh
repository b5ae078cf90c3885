#!/bin/bash
# round 6, thirteenth call: f16 rows on the f16 matrix core -- Gram / build parity tests, then the 1 M x 768 f16 build + search leg (and the widened form beside it)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_build.py tests/test_gpu_tie_order.py tests/test_gpu_sharding.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 600 python bench.py --only large768f16 > $O/large768f16.json 2> $O/large768f16.err; tail -2 $O/large768f16.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06m/large768f16.json').read().strip().splitlines()[-1])['roofline_large']
print('f16 build', d.get('build'), )
print({k:d[k] for k in ('L','frac','achieved','avg_kernel_ms') if k in d})
print(json.dumps(d.get('mfma'))[:1200])
PY

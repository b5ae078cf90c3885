#!/bin/bash
# pair kernel A/B on one box: parity tests, then u8 at L = 26 / 64 (1 M) and the 10 M u8 leg at L = 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r06r}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_visited16.py tests/test_gpu_quant.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for L in 26 64; do timeout 300 python bench.py --only u8 --L $L > $O/u8_L$L.json 2> $O/u8_L$L.err; python - <<PY
import json
d=json.loads(open('$O/u8_L$L.json').read().strip().splitlines()[-1])['u8']
print('u8 L$L', round(d['avg_kernel_ms'],4), 'ms frac', round(d['frac_of_hbm_peak'],4), 'oracle', d['oracle_sample']['ids_identical_to_gpu'])
PY
done
timeout 300 python bench.py --only sq8 --L 26 > $O/sq8.json 2> $O/sq8.err; python - <<PY
import json
d=json.loads(open('$O/sq8.json').read().strip().splitlines()[-1])['sq8']
print('sq8 L26', d['search_kernel']['avg_kernel_ms'], d['search_kernel']['frac_of_hbm_peak'])
PY
timeout 600 python bench.py --only large_u8 --L 64 > $O/large_u8.json 2> $O/large_u8.err; python - <<PY
import json
d=json.loads(open('$O/large_u8.json').read().strip().splitlines()[-1])['roofline_large_u8']['L64']
print('10M u8 L64', round(d['avg_kernel_ms'],4), 'ms frac', round(d['frac'],4), 'oracle', d['oracle_sample']['ids_identical_to_gpu'])
PY

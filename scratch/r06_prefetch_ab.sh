#!/bin/bash
# pair kernel with / without the filtered row prefetch (DANN_TUNE_ON=2), same box: parity with it on, then 1 M u8 at L = 26 / 64, SQ-8 L = 26, 10 M u8 at L = 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r06v}; mkdir -p $O
DANN_TUNE_ON=2 timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_visited16.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for T in 0 2 0 2; do
for L in 26 64; do DANN_TUNE_ON=$T timeout 300 python bench.py --only u8 --L $L > $O/u8_L${L}_t$T.json 2> /dev/null; python - <<PY
import json
d=json.loads(open('$O/u8_L${L}_t$T.json').read().strip().splitlines()[-1])['u8']
print('tune_on=$T u8 1M L$L', round(d['avg_kernel_ms'],4), 'ms frac', round(d['frac_of_hbm_peak'],4), d['oracle_sample']['ids_identical_to_gpu'])
PY
done; done
for T in 0 2; do DANN_TUNE_ON=$T timeout 600 python bench.py --only large_u8 --L 64 > $O/large_u8_t$T.json 2> /dev/null; python - <<PY
import json
d=json.loads(open('$O/large_u8_t$T.json').read().strip().splitlines()[-1])['roofline_large_u8']['L64']
print('tune_on=$T 10M u8 L64', round(d['avg_kernel_ms'],4), 'ms frac', round(d['frac'],4), d['oracle_sample']['ids_identical_to_gpu'])
PY
done

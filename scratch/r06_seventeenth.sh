#!/bin/bash
# round 6, seventeenth call: pair kernel with four queue entries per lane (L + start points <= 128) -- parity, then SQ-8 at 10 M (its L = 96 leg moves to the pair kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_visited16.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --only large_sq8 > $O/large_sq8.json 2> $O/large_sq8.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06t/large_sq8.json').read().strip().splitlines()[-1])['roofline_large_sq8']
for k,v in d.items():
    if k.startswith('L') and isinstance(v,dict): print(k, v['kernel_family'], round(v['avg_kernel_ms'],3), 'frac', round(v['frac'],4), 'recall', v['recall_at_10_vs_exact_f32_no_rerank'], v['oracle_sample']['ids_identical_to_gpu'])
PY
for L in 100 120; do timeout 300 python bench.py --only u8 --L $L > $O/u8_L$L.json 2> $O/u8_L$L.err; python - <<PY
import json
d=json.loads(open('$O/u8_L$L.json').read().strip().splitlines()[-1])['u8']
print('u8 1M L$L', d['kernel_family'], round(d['avg_kernel_ms'],4), 'ms frac', round(d['frac_of_hbm_peak'],4), d['oracle_sample']['ids_identical_to_gpu'])
PY
done

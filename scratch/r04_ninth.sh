#!/bin/bash
# general-size 16-bit tables: parity, then the pair kernel at 7 (auto) and 6 LDS granules per wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pair.py tests/test_gpu_visited16.py tests/test_gpu_parity.py tests/test_gpu_quant.py tests/test_gpu_server.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest.log
DANN_TEST_VISITED_FORMAT=16 DANN_TUNE_OFF=4 timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -6 > $O/pytest_fmt16.log
for w in 0 768 928 1024; do
  vb=""; [ $w != 0 ] && vb="--visited-format 16 --visited-bits $w"
  for wl in u8 sq8; do
  DANN_DEBUG=1 timeout 200 python bench.py --only $wl $vb 2>$O/err_${wl}_$w.log | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('$wl words=$w', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak')}, v.get('oracle_sample'))"
  grep "visited cap" $O/err_${wl}_$w.log | tail -1
  done
done > $O/pair_sizes.txt 2>&1
timeout 300 python bench.py --only large 2>/dev/null | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read())['roofline_large']
print({k:o.get(k) for k in ('avg_kernel_ms','frac','qps','L','recall_at_10','oracle_sample')})" > $O/large.txt 2>&1

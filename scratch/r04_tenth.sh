#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pair.py tests/test_gpu_visited16.py tests/test_gpu_quant.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -6 > $O/pytest.log
for wl in u8 sq8 u8 sq8; do
  timeout 200 python bench.py --only $wl 2>/dev/null | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('$wl', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak')}, v.get('oracle_sample'))"
done > $O/pair.txt 2>&1

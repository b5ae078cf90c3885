#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/u8ab
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quant.py tests/test_gpu_edges.py -m gpu -q --timeout=200 -p no:cacheprovider 2>&1 | tail -1
for w in u8 sq8; do timeout 200 python bench.py --only $w 2>/dev/null | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('$w', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak','algorithmic_GBps','recall_at_10_vs_exact_f32_no_rerank')}, v.get('oracle_sample'))"; done
timeout 300 python bench.py --no-cpu-baseline --no-extras --large none 2>/dev/null | tail -1 | cut -c1-160

#!/bin/bash
# round 4, second GPU call: the pair kernel -- parity, then u8 / sq8 A/B against one wave per query (DANN_TUNE_OFF=16)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair.py -m gpu -q --timeout=400 -p no:cacheprovider -x 2>&1 | tail -25 > $O/pytest_pair.log
export DANN_DEBUG=1
one() { # tune workload
  DANN_TUNE_OFF=$1 timeout 200 python bench.py --only $2 2>$O/err_$1_$2.log | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('tune_off=$1 $2', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak','algorithmic_GBps','recall_at_10_vs_exact_f32_no_rerank')}, v.get('oracle_sample'))"
  grep "visited cap" $O/err_$1_$2.log | tail -1
}
for w in u8 sq8; do one 16 $w; one 0 $w; one 16 $w; one 0 $w; done > $O/ab_pair.log 2>&1
unset DANN_DEBUG
timeout 900 python -m pytest tests/test_gpu_quant.py tests/test_gpu_parity.py tests/test_gpu_server.py tests/test_gpu_visited16.py -m gpu -q --timeout=400 -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_some.log

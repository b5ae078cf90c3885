#!/bin/bash
# final pair kernel of the round against the round-5 one (libdann_base.so), same box, interleaved + the pair / visited / quant parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r06aa}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_visited16.py tests/test_gpu_quant.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
run() { n=$1; lib=$R/diskann_amd/libdann_$2.so; shift 2
  DANN_LIB_PATH=$lib timeout 600 python bench.py "$@" > $O/$n.json 2> /dev/null
  python - <<PY
import json
d=list(json.loads(open('$O/$n.json').read().strip().splitlines()[-1]).values())[0]
if 'L64' in d and 'avg_kernel_ms' not in d: d=d['L64']
print('$n', round(d['avg_kernel_ms'],4), 'ms', d.get('oracle_sample',{}).get('ids_identical_to_gpu'))
PY
}
for rep in 1 2 3; do for v in base hip; do run ${v}_u8_L26_$rep $v --only u8 --L 26; run ${v}_sq8_L26_$rep $v --only sq8 --L 26; run ${v}_u8_L64_$rep $v --only u8 --L 64; done; done
for rep in 1 2; do for v in base hip; do run ${v}_large_u8_$rep $v --only large_u8 --L 64; done; done

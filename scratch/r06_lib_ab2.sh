#!/bin/bash
# same-box A/B of pair-kernel variants at QE = 3 (u8, L = 64, 1 M and 10 M): va = k-ary + scalar survivor loop, vb = binary + vector loop, vc = binary + scalar loop (QE >= 2 only; QE = 1 always k-ary + vector), kary = both
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r06y}; mkdir -p $O
run() { n=$1; lib=$R/diskann_amd/libdann_$2.so; shift 2
  DANN_LIB_PATH=$lib timeout 600 python bench.py "$@" > $O/$n.json 2> /dev/null
  python - <<PY
import json
d=list(json.loads(open('$O/$n.json').read().strip().splitlines()[-1]).values())[0]
if 'L64' in d and 'avg_kernel_ms' not in d: d=d['L64']
print('$n', round(d['avg_kernel_ms'],4), 'ms', d.get('oracle_sample',{}).get('ids_identical_to_gpu'))
PY
}
for rep in 1 2 3; do for v in base kary va vb vc; do run ${v}_u8_L64_$rep $v --only u8 --L 64; done; done
for rep in 1 2; do for v in base kary va vb vc; do run ${v}_u8_L40_$rep $v --only u8 --L 40; done; done
for v in base kary va vb vc; do run ${v}_large_u8 $v --only large_u8 --L 64; done

#!/bin/bash
# same-box A/B of two builds of the library (DANN_LIB_PATH), interleaved: u8 at L = 26 / 64 (1 M), SQ-8 at L = 26, 10 M u8 at L = 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r06x}; mkdir -p $O
A=$R/diskann_amd/libdann_base.so; B=$R/diskann_amd/libdann_kary.so
run() { # name lib args...
  n=$1; lib=$2; shift 2
  DANN_LIB_PATH=$lib timeout 600 python bench.py "$@" > $O/$n.json 2> /dev/null
  python - <<PY
import json
d=list(json.loads(open('$O/$n.json').read().strip().splitlines()[-1]).values())[0]
if 'L64' in d and 'avg_kernel_ms' not in d: d=d['L64']
print('$n', round(d['avg_kernel_ms'],4), 'ms')
PY
}
for rep in 1 2 3; do
  run base_u8_L26_$rep $A --only u8 --L 26; run kary_u8_L26_$rep $B --only u8 --L 26
  run base_u8_L64_$rep $A --only u8 --L 64; run kary_u8_L64_$rep $B --only u8 --L 64
  run base_sq8_L26_$rep $A --only sq8 --L 26; run kary_sq8_L26_$rep $B --only sq8 --L 26
done
for rep in 1 2; do run base_large_u8_$rep $A --only large_u8 --L 64; run kary_large_u8_$rep $B --only large_u8 --L 64; done

#!/bin/sh
# A/B of scratch/bin/libdann_<v>.so variants: latency lab (tune 0), then the u8 search-only workload
for v in base adjmerge base adjmerge; do
  export DANN_LIB_PATH=$PWD/scratch/bin/libdann_$v.so
  echo "=== $v"; timeout 200 python scratch/latency_lab.py --tunes 0 2>&1 | grep -v "amdgpu.ids\|DANN_TUNE"
done
for v in base adjmerge base adjmerge; do
  export DANN_LIB_PATH=$PWD/scratch/bin/libdann_$v.so
  echo "=== u8 $v"; timeout 100 python bench.py --only u8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['u8']; print(d['avg_kernel_ms'], d['qps'])"
done

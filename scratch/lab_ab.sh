#!/bin/sh
# A/B: scratch/bin/libdann_base.so vs the in-tree library on the u8 / sq8 search-only workloads; then the parity tests
for v in ${VARIANTS:-base new base new}; do
  if [ $v = new ]; then unset DANN_LIB_PATH; else export DANN_LIB_PATH=$PWD/scratch/bin/libdann_$v.so; fi
  for w in u8 sq8; do
  echo "=== $w $v"; timeout 100 python bench.py --only $w 2>/dev/null | python -c "import sys,json; d=list(json.loads(sys.stdin.read().strip().splitlines()[-1]).values())[0]; print(d['avg_kernel_ms'], d['qps'])"
  done
done
unset DANN_LIB_PATH
timeout 400 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -4

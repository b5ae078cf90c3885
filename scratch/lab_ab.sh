#!/bin/sh
# A/B of scratch/bin/libdann_<v>.so variants (new = the in-tree library): latency lab (tune 0), u8/sq8, parity tests
for v in ${VARIANTS:-base new}; do
  if [ $v = new ]; then unset DANN_LIB_PATH; else export DANN_LIB_PATH=$PWD/scratch/bin/libdann_$v.so; fi
  echo "=== $v"; timeout 200 python scratch/latency_lab.py --tunes 0 2>&1 | grep -v "amdgpu.ids\|DANN_TUNE\|max_concurrency 256"
  for w in ${WORKLOADS:-u8}; do
  echo "--- $w $v"; timeout 100 python bench.py --only $w 2>/dev/null | python -c "import sys,json; d=list(json.loads(sys.stdin.read().strip().splitlines()[-1]).values())[0]; print(d['avg_kernel_ms'], d['qps'])"
  done
done
unset DANN_LIB_PATH
if [ -z "${NO_TESTS:-}" ]; then timeout 400 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -4; fi

#!/bin/sh
# A/B of libdann variants (scratch/bin/libdann_<v>.so vs the in-tree build) on the 1 M x 768 f16 index
mkdir -p gpurun_out
for v in default lazy default lazy; do
  if [ $v = default ]; then unset DANN_LIB_PATH; else export DANN_LIB_PATH=$PWD/scratch/bin/libdann_$v.so; fi
  timeout 150 python bench.py --only large768f16 --graph-cache /tmp/g768f16 2> gpurun_out/f16ab_$v.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline_large']
print('$v', d['avg_kernel_ms'], d['achieved'], d['oracle_sample'])"
done

#!/bin/sh
# 10 M x 128 index at L = 56: the auto-sized visited table vs explicit sizes (occupancy vs spill rate)
for vb in ${SIZES:-0 2048 2560 0}; do
  echo "=== visited entries $vb"
  DANN_DEBUG=1 timeout 200 python bench.py --only large --L 56 --graph-cache /tmp/glarge --visited-bits $vb 2> /tmp/err_$vb.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline_large']
print(d.get('avg_kernel_ms'), d.get('achieved'), d.get('recall_at_10'))"
  grep "L=56" /tmp/err_$vb.txt | tail -1
done

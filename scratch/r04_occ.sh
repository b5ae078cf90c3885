#!/bin/bash
# pair kernel: throughput against resident wavefronts per CU (explicit tables of 1024 / 2048 / 4096 words per query)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04m; mkdir -p $O
for w in 1024 2048 4096; do
  DANN_DEBUG=1 timeout 200 python bench.py --only u8 --visited-format 16 --visited-bits $w 2>$O/err_$w.log | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('words=$w', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak')})"
done > $O/occ.txt 2>&1

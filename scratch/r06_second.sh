#!/bin/bash
# round 6, second call: the 10 M integer legs on their own + the MFMA exactness probe
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06b; mkdir -p $O /tmp/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scratch/probe_mfma_exact.hip -o /tmp/bin/probe_mfma_exact 2> /dev/null
timeout 300 /tmp/bin/probe_mfma_exact > $O/probe_mfma_exact.txt 2>&1
cat $O/probe_mfma_exact.txt | head -40
for K in u8 sq8; do
  timeout 900 python bench.py --only large_$K > $O/large_$K.json 2> $O/large_$K.err
  tail -c 1500 $O/large_$K.json; tail -3 $O/large_$K.err
done

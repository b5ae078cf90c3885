#!/bin/sh
# A/B of scratch/bin/libdann_<v>.so variants (new = the in-tree library): latency lab (tune 0)
for v in ${VARIANTS:-base new}; do
  if [ $v = new ]; then unset DANN_LIB_PATH; else export DANN_LIB_PATH=$PWD/scratch/bin/libdann_$v.so; fi
  echo "=== $v"; timeout 200 python scratch/latency_lab.py --tunes 0 2>&1 | grep -v "amdgpu.ids\|DANN_TUNE"
done

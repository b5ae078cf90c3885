#!/bin/bash
# build scratch/bin/libdann_<name>.so: the in-tree objects with the search translation units recompiled under extra
# flags (A/B builds for scratch/lab_ab.sh, loaded through DANN_LIB_PATH).  usage: build_variant.sh <name> [flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
obj=/tmp/dann_variant_$name; mkdir -p $obj scratch/bin
cp diskann_amd/build/*.o $obj/
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero"
for t in ${UNITS:-search_f32 search_f16 search_u8 search_i8 search_sq8 search_pq}; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c diskann_amd/csrc/$t.hip -o $obj/$t.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/bin/libdann_$name.so $obj/*.o
ls -la scratch/bin/libdann_$name.so

#!/bin/sh
# profiling variant of the library (per-phase cycle counters in the beam-search kernel): diskann_amd/libdann_prof.so
cd "$(dirname "$0")/.." || exit 1
SRC=diskann_amd/csrc
OUT=diskann_amd/build_prof
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -DDANN_PHASE_CYCLES"
for f in api server sharded search_kernels search_f32 search_f16 search_u8 search_i8 search_sq8 search_pq search_pqlut search_pqlut2 search_pqlut3 search_pqlut4 search_pair paged_kernels distance_kernels build_kernels pq_kernels; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diskann_amd/libdann_prof.so $OUT/*.o

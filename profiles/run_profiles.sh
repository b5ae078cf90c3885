#!/bin/bash
# Run on the GPU box (through gpurun): plain bench, kernel trace at the chosen L, separate PMC passes.
# Usage: profiles/run_profiles.sh <tag>     -> gpurun_out/<tag>_*.{json,csv}
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
if [ -z "${SKIP_PLAIN:-}" ]; then python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; fi
if [ -n "${FIXED_L:-}" ]; then L=$FIXED_L; else L=$(python -c "import json;print(json.load(open('$OUT/${TAG}_bench.json'))['config']['L'])"); fi
# kernel trace with L fixed: every beam_search_kernel launch of this run is the timed workload
if [ -z "${SKIP_TRACE:-}" ]; then
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG/trace -o t -- python $R/bench.py --steps 20 --no-cpu-baseline --no-extras --L $L --graph-cache /tmp/prof_$TAG/graph.bin \
    > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof_$TAG/trace.err
python $R/profiles/summarize_rocprof.py trace /tmp/prof_$TAG/trace/t_results.db $OUT/${TAG}_kernel_trace.csv 20
fi
if [ ! -f /tmp/prof_$TAG/graph.bin ]; then  # the PMC passes load the graph: no build dispatches under the counters
    python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --L $L --graph-cache /tmp/prof_$TAG/graph.bin > /dev/null 2>&1
fi
if [ -n "${SKIP_PMC:-}" ]; then exit 0; fi
# PMC_SHORT=1: the traffic counters and the instruction mix only (4 passes instead of 6)
if [ -n "${PMC_SHORT:-}" ]; then
  set -- FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
else
  set -- FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"
fi
for C in "$@"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace -d /tmp/prof_$TAG/pmc_$N -o p -- python $R/bench.py --steps 3 --warmup 1 \
        --no-cpu-baseline --no-extras --L $L --graph-cache /tmp/prof_$TAG/graph.bin > /dev/null 2> /tmp/prof_$TAG/pmc_$N.err
    python $R/profiles/summarize_rocprof.py pmc /tmp/prof_$TAG/pmc_$N/p_results.db $OUT/${TAG}_pmc_$N.csv beam_search
done
ls -la $OUT | tail -12

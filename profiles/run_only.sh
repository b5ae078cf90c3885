#!/bin/bash
# One secondary workload of bench.py (--only <workload>) under rocprofv3, on the GPU box (through gpurun):
#   plain run -> kernel trace + stats -> separate PMC passes (FETCH_SIZE | WRITE_SIZE | TCC hit/miss), as the
#   MI355X guide prescribes (counters in their own runs, --kernel-trace only).
# Usage: profiles/run_only.sh <tag> <workload> [bench args...]   -> gpurun_out/<tag>_<workload>_*.{json,csv}
set -u
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=/tmp/prof_${TAG}_$WL
rm -rf $D && mkdir -p $D
case $WL in gather) FLT=expand_beam;; *) FLT=search_kernel;; esac
python $R/bench.py --only $WL "$@" > $OUT/${TAG}_${WL}.json 2> $OUT/${TAG}_${WL}.err
L=$(python -c "import json;d=json.loads(open('$OUT/${TAG}_${WL}.json').read().strip().splitlines()[-1]);v=list(d.values())[0];print(v.get('L',0))")
LARG=""; if [ "$L" != "0" ]; then LARG="--L $L"; fi
timeout 400 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python $R/bench.py --only $WL $LARG "$@" \
    > $OUT/${TAG}_${WL}_under_rocprof.json 2> $D/trace.err
python $R/profiles/summarize_rocprof.py trace $D/trace/t_results.db $OUT/${TAG}_${WL}_kernel_trace.csv 12
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 400 rocprofv3 --pmc $C --kernel-trace -d $D/pmc_$N -o p -- python $R/bench.py --only $WL $LARG "$@" > /dev/null 2> $D/pmc_$N.err
    python $R/profiles/summarize_rocprof.py pmc $D/pmc_$N/p_results.db $OUT/${TAG}_${WL}_pmc_$N.csv $FLT
done

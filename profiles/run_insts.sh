#!/bin/bash
# Instruction mix and issue/wait cycles of one bench.py --only workload (two PMC passes, --kernel-trace only).
# Usage: profiles/run_insts.sh <tag> <workload> [bench args...]  -> gpurun_out/<tag>_<workload>_insts_{1,2}.csv
set -u
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=/tmp/insts_${TAG}_$WL
rm -rf $D && mkdir -p $D
case $WL in gather) FLT=expand_beam;; *) FLT=beam_search;; esac
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --kernel-trace -d $D/p$i -o p -- python $R/bench.py --only $WL "$@" > /dev/null 2> $D/p$i.err
    python $R/profiles/summarize_rocprof.py pmc $D/p$i/p_results.db $OUT/${TAG}_${WL}_insts_$i.csv $FLT || tail -5 $D/p$i.err
done

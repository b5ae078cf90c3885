#!/bin/bash
# instruction mix / wave-state / LDS / traffic counters of the PQ search kernel (separate passes, kernel trace only)
# usage: scratch/r05_pq_pmc.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05p}; shift
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); D=/tmp/pm_pq_$i; rm -rf $D
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/bench.py --only pq --L 96 "$@" > /dev/null 2> $O/err_$i.log
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/pq_pmc_$i.csv search > /dev/null 2>&1
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/pq_pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        if int(r["grid_size"]) >= 3000000: print(f.split("/")[-1], r["kernel"][24:64], r["grid_size"], r["lds_bytes"], r["vgprs"], r["sgprs"], r["counter"], r["avg_value"], r["avg_duration_us"])
PY

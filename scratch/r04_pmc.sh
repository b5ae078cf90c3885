#!/bin/bash
# instruction and wave-state counters of the u8 search, one wave per query (DANN_TUNE_OFF=16) vs two queries per wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04c}; WL=${2:-u8}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 16 0; do
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1)); D=/tmp/pm_${mode}_$i; rm -rf $D
    DANN_TUNE_OFF=$mode timeout 300 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/bench.py --only $WL --L 26 > /dev/null 2> $O/err_${mode}_$i.log
    python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/${WL}_tune${mode}_pmc_$i.csv search > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/${WL}_tune*_pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        if int(r["grid_size"]) >= 3000000: print(f.split("/")[-1], r["kernel"][30:70], r["grid_size"], r["lds_bytes"], r["vgprs"], r["sgprs"], r["counter"], r["avg_value"], r["avg_duration_us"])
PY

#!/bin/bash
# counters of gram_tiles_kernel (wide and narrow instantiation) on the 1 M x 768 build
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04x}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="1000000 768 64 56 128 16384"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1)); D=/tmp/pmb_$i; rm -rf $D
  timeout 500 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/scratch/build_phases.py $A > /dev/null 2> $O/err_$i.log
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/gram_tiles_pmc_$i.csv gram_tiles > /dev/null 2>&1
  tail -3 $O/err_$i.log
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/gram_tiles_pmc_*.csv")):
    best = {}
    for r in csv.DictReader(open(f)):
        k = (r["kernel"][45:75], r["counter"])
        if k not in best or float(r["avg_duration_us"]) * int(r["dispatches"]) > float(best[k]["avg_duration_us"]) * int(best[k]["dispatches"]): best[k] = r
    for k, r in sorted(best.items()):
        print(k[0], r["grid_size"], r["lds_bytes"], r["vgprs"], r["dispatches"], r["counter"], r["avg_value"], r["avg_duration_us"])
PY

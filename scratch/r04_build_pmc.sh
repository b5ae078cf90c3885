#!/bin/bash
# NOTE: the DANN_BUILD_ONE_STREAM / DANN_BUILD_ITEM_ORDER / DANN_GRAM_ONE_KERNEL switches this A/B used were removed from the library after the
# measurement (results: profiles/r04q_*); check out commit 1e8b133 to repeat it.
# 1 M x 768 f32 build: (a) long back-edge lists on the side stream vs one stream, wall clock, three runs each;
# (b) kernel trace; (c) instruction / wave-state counters of the prune kernels (pool_sweep, gram_tiles, backedge)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04q}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="1000000 768 64 56 128 16384"
for i in 1 2 3; do
  echo "side stream:" $(timeout 200 python $R/scratch/build_phases.py $A 2>/dev/null | grep -i "build" | head -2 | tr '\n' ' ')
  echo "one stream: " $(DANN_BUILD_ONE_STREAM=1 timeout 200 python $R/scratch/build_phases.py $A 2>/dev/null | grep -i "build" | head -2 | tr '\n' ' ')
done > $O/ab.txt 2>&1
cat $O/ab.txt
rm -rf /tmp/pm && DANN_DEBUG=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $O/build768.log 2> $O/build768.err
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $O/build768_kernel_trace.csv 14 > /dev/null 2>&1
python $R/profiles/condense_build.py $O/build768.log $O/build768_kernel_trace.csv $O/build768_summary.json > /dev/null 2>&1
python - <<PY
import json
o=json.load(open("$O/build768_summary.json"))
print({k:(round(v["total_ms"]), v["calls"]) for k,v in o["kernels"].items()}); print(o["model"]["build_seconds"])
PY
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); D=/tmp/pmb_$i; rm -rf $D
  timeout 500 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/scratch/build_phases.py $A > /dev/null 2> $O/err_$i.log
  for K in pool_sweep gram_tiles backedge_kernel pool_sort backedge_list; do
    python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/${K}_pmc_$i.csv $K > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/*_pmc_*.csv")):
    best = {}
    for r in csv.DictReader(open(f)):
        k = (r["counter"])
        if k not in best or float(r["avg_duration_us"]) * int(r["dispatches"]) > float(best[k]["avg_duration_us"]) * int(best[k]["dispatches"]): best[k] = r
    for r in best.values():
        print(f.split("/")[-1], r["grid_size"], r["lds_bytes"], r["vgprs"], r["sgprs"], r["dispatches"], r["counter"], r["avg_value"], r["avg_duration_us"])
PY

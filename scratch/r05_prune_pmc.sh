#!/bin/bash
# counters of the prune kernels inside the 10 M x 128 build (`bench.py --only large --L 56`)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05x2}
O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); D=/tmp/pm_prune_$i; rm -rf $D
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/bench.py --only large --L 56 > /dev/null 2> $O/err_$i.log
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/prune_pmc_$i.csv prune > /dev/null 2>&1
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/backedge_pmc_$i.csv backedge > /dev/null 2>&1
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/*_pmc_*.csv")):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["avg_duration_us"])*int(r["dispatches"]))
    seen=set()
    for r in rows[:60]:
        key=(r["kernel"][:60],r["grid_size"],r["lds_bytes"])
        if len(seen)>=4 and key not in seen: continue
        seen.add(key)
        print(f.split("/")[-1], r["kernel"].split("::")[-1][:28], "grid",r["grid_size"],"lds",r["lds_bytes"],"vgpr",r["vgprs"],r["counter"],r["dispatches"],r["avg_value"],r["avg_duration_us"])
PY

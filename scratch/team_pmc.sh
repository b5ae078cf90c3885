#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r03z}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp && timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES --kernel-trace -d /tmp/tp/p -o p -- python $R/scratch/team_trace.py > $R/gpurun_out/${T}_team_pmc.log 2>&1
python $R/profiles/summarize_rocprof.py pmc /tmp/tp/p/p_results.db $R/gpurun_out/${T}_team_pmc.csv beam_search > /dev/null 2>&1
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/${T}_team_pmc.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if ", 4>" in r["kernel"]: agg[(r["kernel"].split("beam_search_kernel")[1][:30], r["grid_size"], r["counter"])].append(float(r["avg_value"]))
for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)), len(v))
PY

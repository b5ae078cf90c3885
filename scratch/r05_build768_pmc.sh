#!/bin/bash
# counters of the insert-time search inside the 1 M x 768 f32 build
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05b768p}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SPEC=${2:-1000000:768:64:56:128:f32}
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  i=$((i+1)); D=/tmp/pm768_$i; rm -rf $D
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/bench.py --only build768 --build-spec $SPEC > /dev/null 2> $O/err_$i.log
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/search_pmc_$i.csv beam_search > /dev/null 2>&1
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/search_pmc_*.csv")):
    rows=[r for r in csv.DictReader(open(f)) if "4, 0, 0, 0, 1, true" in r["kernel"]]
    agg={}
    for r in rows:
        k=(r["lds_bytes"],r["vgprs"],r["counter"]); e=agg.setdefault(k,[0,0.0,0.0,0])
        n=int(r["dispatches"]); e[0]+=n; e[1]+=float(r["avg_value"])*n; e[2]+=float(r["avg_duration_us"])*n; e[3]+=int(r["grid_size"])*n
    for k,e in sorted(agg.items(), key=lambda kv:-kv[1][2])[:24]:
        print(f.split("/")[-1], "lds",k[0],"vgpr",k[1],k[2],"launches",e[0],"sum",int(e[1]),"sum_us",int(e[2]),"threads",e[3])
PY

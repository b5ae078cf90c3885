#!/bin/bash
# Round-5 evidence on one box: parity suite + smoke, default bench line, headline trace + PMC, u8 / sq8 / pq passes,
# instruction / LDS counters of the PQ and pair kernels.  usage: scratch/r05_final.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05z}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3 | tee gpurun_out/${T}_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/${T}_pytest.log
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -2 gpurun_out/${T}_bench.err
cut -c1-300 gpurun_out/${T}_bench.json
SKIP_PLAIN=1 PMC_SHORT=1 timeout 500 bash profiles/run_profiles.sh $T > gpurun_out/${T}_profiles.log 2>&1; tail -3 gpurun_out/${T}_profiles.log
for w in u8 sq8 pq; do timeout 300 bash profiles/run_only.sh $T $w > gpurun_out/${T}_only_$w.log 2>&1; done
timeout 300 bash profiles/run_only.sh ${T}L64 u8 --L 64 > gpurun_out/${T}_only_u8_L64.log 2>&1
timeout 400 bash scratch/r05_pq_pmc.sh ${T}_pqpmc > gpurun_out/${T}_pqpmc.log 2>&1; tail -30 gpurun_out/${T}_pqpmc.log
# pair kernel counters (u8, L = 26 and L = 64)
O=$R/gpurun_out/${T}_pairpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in 26 64; do i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); D=/tmp/pm_pair_${L}_$i; rm -rf $D
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/bench.py --only u8 --L $L > /dev/null 2> $O/err_${L}_$i.log
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/u8_L${L}_pmc_$i.csv search > /dev/null 2>&1
done; done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/u8_L*_pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        if int(r["grid_size"]) >= 3000000 and int(r["dispatches"]) >= 5: print(f.split("/")[-1], r["kernel"][24:70], r["grid_size"], r["lds_bytes"], r["vgprs"], r["counter"], r["avg_value"], r["avg_duration_us"])
PY
ls $R/gpurun_out | grep $T | wc -l

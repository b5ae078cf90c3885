# kernel trace + MFMA counters of the 1 M x 768 build with the MFMA back-edge path (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
A="${2:-1000000 768 64 56 128 16384 --mfma}"
rm -rf /tmp/pm && rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $R/gpurun_out/${TAG}_mfma_build.log 2>&1
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $R/gpurun_out/${TAG}_mfma_build_kernel_trace.csv 10
for C in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace -d /tmp/pm/p_$N -o p -- python $R/scratch/build_phases.py $A > /dev/null 2>&1
    python $R/profiles/summarize_rocprof.py pmc /tmp/pm/p_$N/p_results.db $R/gpurun_out/${TAG}_mfma_build_pmc_$N.csv backedge_gram
done
cat $R/gpurun_out/${TAG}_mfma_build.log | tail -2

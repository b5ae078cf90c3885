cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pb && rocprofv3 --kernel-trace --stats -d /tmp/pb -o t -- python $R/scratch/build_phases.py "$@" > /tmp/pb.log 2>&1
grep max_batch /tmp/pb.log
python $R/profiles/summarize_rocprof.py trace /tmp/pb/t_results.db $R/gpurun_out/build_trace.csv 14
cut -c1-90,91- $R/gpurun_out/build_trace.csv | awk -F'",' '{print substr($1,1,70), $2}' | head -16

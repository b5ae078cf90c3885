#!/bin/bash
# NOTE: the DANN_BUILD_ONE_STREAM / DANN_BUILD_ITEM_ORDER / DANN_GRAM_ONE_KERNEL switches this A/B used were removed from the library after the
# measurement (results: profiles/r04q_*); check out commit 1e8b133 to repeat it.
# 1 M x 768 f32 build wall clock: default vs a development switch given as $2 (env assignment), three runs each;
# then the build tests and a kernel trace of the default
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04r}; SW=${2:-DANN_BUILD_ITEM_ORDER=0}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A=${3:-"1000000 768 64 56 128 16384"}
for i in 1 2 3; do
  echo "default:" $(timeout 200 python $R/scratch/build_phases.py $A 2>/dev/null | grep -o "build [0-9.]*s" | head -1)
  echo "$SW:" $(env $SW timeout 200 python $R/scratch/build_phases.py $A 2>/dev/null | grep -o "build [0-9.]*s" | head -1)
done > $O/ab.txt 2>&1
cat $O/ab.txt
if [ -z "$NO_TESTS" ]; then (cd $R && timeout 900 python -m pytest tests/test_gpu_build.py -x -q 2>&1 | tail -3) | tee $O/pytest_build.log; fi
rm -rf /tmp/pm && DANN_DEBUG=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pm/t -o t -- python $R/scratch/build_phases.py $A > $O/build768.log 2> $O/build768.err
python $R/profiles/summarize_rocprof.py trace /tmp/pm/t/t_results.db $O/build768_kernel_trace.csv 14 > /dev/null 2>&1
python $R/profiles/condense_build.py $O/build768.log $O/build768_kernel_trace.csv $O/build768_summary.json > /dev/null 2>&1
python - <<PY
import json
o=json.load(open("$O/build768_summary.json"))
print({k:(round(v["total_ms"]), v["calls"]) for k,v in o["kernels"].items()}); print(o["model"]["build_seconds"]); print(o["rates"].get("gram_tiles_kernel"))
PY

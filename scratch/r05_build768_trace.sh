#!/bin/bash
# kernel trace of the 1 M x 768 f32 build (bench.py --only build768) with the round's final build kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05b768}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SPEC=${2:-1000000:768:64:56:128:f32}
rm -rf /tmp/kt768
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt768 -o k -- python $R/bench.py --only build768 --build-spec $SPEC > $O/build768.json 2> $O/build768.err
python $R/profiles/summarize_rocprof.py trace /tmp/kt768/k_results.db $O/build768_kernel_trace.csv 20 > /dev/null 2>&1
cut -c1-170 $O/build768_kernel_trace.csv
grep -E "build [0-9.]+s" $O/build768.err | tail -2
[ -n "$NOPLAIN" ] || timeout 300 python $R/bench.py --only build768 --build-spec $SPEC > $O/build768_plain.json 2> $O/build768_plain.err
grep -E "build [0-9.]+s" $O/build768_plain.err | tail -2

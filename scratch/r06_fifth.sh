#!/bin/bash
# round 6, fifth call: where the PQ trainer's 3 s go (kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06e; mkdir -p $O
export PYTHONPATH=$R
python scratch/r06_train_time.py > $O/train_time.txt 2>&1; cat $O/train_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- python $R/scratch/r06_train_time.py > /dev/null 2>&1
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1); cp $f $O/train_kernel_stats.csv; head -25 $f | cut -c1-200

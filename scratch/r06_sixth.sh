#!/bin/bash
# round 6, sixth call: the reworked PQ trainer -- parity tests, then its time at the bench's shape (pivot digest of the old kernels: 4bc82ec22959501c / seeds 26f1a0ca69380ab7)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06f; mkdir -p $O
export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_quant.py -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python scratch/r06_train_time.py > $O/train_time.txt 2>&1; cat $O/train_time.txt

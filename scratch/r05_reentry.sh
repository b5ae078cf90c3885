#!/bin/bash
# Round-5 re-entry check on one box: the tie-order tests, the whole GPU suite + smoke, the default bench line.
# usage: scratch/r05_reentry.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05y}
cd $R; mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_tie_order.py -q --timeout 200 --tb=short 2>&1 | tail -60 > gpurun_out/${T}_tie.log; tail -25 gpurun_out/${T}_tie.log
timeout 400 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_tie_order.py 2>&1 | tail -15 | tee gpurun_out/${T}_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/${T}_pytest.log
t0=$(date +%s)
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/${T}_bench.err
cut -c1-400 gpurun_out/${T}_bench.json

#!/bin/bash
# Round-5 re-entry, final check of the tree on one box: GPU suite + smoke, default bench line, tie-order cost.
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r05zz}
cd $R; mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -15 | tee gpurun_out/${T}_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/${T}_pytest.log
timeout 150 python scratch/tie_order_cost.py 200000 > gpurun_out/${T}_tie_cost.json 2> gpurun_out/${T}_tie_cost.err; cat gpurun_out/${T}_tie_cost.json
t0=$(date +%s)
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/${T}_bench.err
cut -c1-300 gpurun_out/${T}_bench.json

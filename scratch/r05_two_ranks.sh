#!/bin/bash
# the N > 1 code path of bench.py with all extras, two ranks on the one GPU of the box (gloo; format / correctness, not a scaling number)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
T=${1:-r05q}
t0=$(date +%s)
DANN_BENCH_ONE_DEVICE=1 OMP_NUM_THREADS=1 timeout 1200 python bench.py --gpus 2 > gpurun_out/${T}_bench_2ranks.json 2> gpurun_out/${T}_bench_2ranks.err; echo "rc=$? wall $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/${T}_bench_2ranks.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${T}_bench_2ranks.json").read().splitlines() if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "value", round(d["value"]), "scaling", d["scaling"], "build_exchange", d["config"].get("build_exchange"))
print("strong", d["other_configs"]["strong_scaling_shared_set"])
print("legs", [k for k in d["other_configs"]], "large" , "roofline_large" in d)
PY
DANN_TEST_VISITED_FORMAT=16 DANN_TUNE_OFF=4 timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -3

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
timeout 300 python scratch/team_lab.py 1000000 > $OUT/team_lab.log 2>&1
grep -v "^{" $OUT/team_lab.log | tail -30
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
tail -5 $OUT/bench.err
python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/r03b/bench.json")
try:
    o=json.load(open(p))
    print("value",o["value"],"roofline",o["roofline"]["frac"])
    oc=o["other_configs"]
    for k in oc:
        if isinstance(oc[k],(int,float,bool,type(None))): print(k,oc[k])
    print(json.dumps(oc.get("concurrent_callers"),indent=0)[:3000])
    print(json.dumps(oc.get("single_query_L64_latency")), json.dumps(oc.get("batch_L64")), json.dumps(oc.get("host_pointer_search_batch")))
    print(json.dumps(oc.get("cpu_single_query_L64")), json.dumps(oc.get("cpu_1024_queries_at_L")))
except Exception as e:
    print("bench parse failed",e)
PY

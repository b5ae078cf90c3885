#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python scratch/team_repro.py 2>&1 | grep -E "BAD|tune" | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=200 -p no:cacheprovider -k team 2>&1 | tail -1
timeout 300 python - <<'PY'
import os, sys
sys.argv=['x']
exec(open('scratch/prefetch_ab.py').read().split("for rnd in range(2):")[0])
for rnd in range(2):
    for tune in ("0",):
        os.environ["DANN_TUNE_OFF"] = tune
        print(f"tune_off {tune}: " + "  ".join(f"{nq}x{L}: {timed(nq, L, 200 if nq < 64 else 60):.1f} us" for nq, L in ((1, 64), (1, 26), (16, 64), (256, 26), (1024, 26))), flush=True)
PY

#!/bin/bash
# Round-3 evidence at the head (search side): parity suite + smoke, default bench line, headline trace + PMC, u8 / sq8
# profiles, the team A/B lab and the per-phase cycle counters of a team.  (Build-side evidence: scratch/r03_final.sh.)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r03z}
cd $R; mkdir -p gpurun_out
bash scratch/final_run.sh $T
timeout 400 python scratch/team_lab.py 1000000 > gpurun_out/${T}_team_lab.log 2>&1
grep -v "^{" gpurun_out/${T}_team_lab.log | grep -v "^/opt" | tail -21
timeout 300 python scratch/latency_lab.py --prof --tunes 0 --points 1:64:300,1:26:300,1024:26:50 > gpurun_out/${T}_team_phases.log 2>&1
grep -v "^/opt" gpurun_out/${T}_team_phases.log | tail -7

#!/bin/bash
# HBM-side traffic of the build's dominant kernel (the insert search) on the 1 M x 768 f32 build: FETCH_SIZE / WRITE_SIZE passes
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04zv}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="1000000 768 64 56 128 16384"
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmt_$C; rm -rf $D
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $D -o p -- python $R/scratch/build_phases.py $A > $O/build_$C.log 2> $O/err_$C.log
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/insert_search_pmc_$C.csv beam_search > /dev/null 2>&1
  python $R/profiles/summarize_rocprof.py pmc $D/p_results.db $O/gram_tiles_pmc_$C.csv gram_tiles > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json
tot = {}
for f in sorted(glob.glob("$O/*_pmc_*_SIZE.csv")):
    k = f.split("/")[-1].split("_pmc_")[0]; c = f.split("_pmc_")[1][:-4]
    s = sum(float(r["avg_value"]) * int(r["dispatches"]) for r in csv.DictReader(open(f)))
    t = sum(float(r["avg_duration_us"]) * int(r["dispatches"]) for r in csv.DictReader(open(f)))
    tot[(k, c)] = (s, t)
    print(k, c, "sum KiB", s, "sum us", t)
log = [l for l in open("$O/build_FETCH_SIZE.log") if l.startswith("{")][-1]
m = json.loads(log)
alg = m["search"]["algorithmic_bytes"]
hbm = (tot[("insert_search", "FETCH_SIZE")][0] * 2 + tot[("insert_search", "WRITE_SIZE")][0]) * 1024
print("insert search: algorithmic", alg, "hbm-side (FETCH x2 + WRITE)", hbm, "ratio", hbm / alg)
g = m["mfma"]["row_bytes_read"]
hg = (tot[("gram_tiles", "FETCH_SIZE")][0] * 2 + tot[("gram_tiles", "WRITE_SIZE")][0]) * 1024
print("gram tiles: rows read", g, "hbm-side", hg, "ratio", hg / g)
PY

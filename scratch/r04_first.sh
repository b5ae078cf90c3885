#!/bin/bash
# round 4, first GPU call: parity of the 16-bit visited table, then A/B of table format x SGPR cap on u8 / sq8 / f32 / 10M
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04a; mkdir -p $O
export DANN_DEBUG=1
timeout 600 python -m pytest tests/test_gpu_visited16.py -m gpu -q --timeout=300 -p no:cacheprovider 2>&1 | grep -v "^\[dann\]" | tail -15 > $O/pytest_v16.log
DANN_TEST_VISITED_FORMAT=16 DANN_TUNE_OFF=4 timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider -x 2>&1 | grep -v "^\[dann\]" | tail -15 > $O/pytest_all_fmt16.log
sel() { # lib: the directory to run bench.py from + the library
  cd $R; unset DANN_LIB_PATH
  case $1 in
    new) ;;
    base) cd $R/scratch/bin/base_tree ;;
    *) export DANN_LIB_PATH=$R/scratch/bin/libdann_$1.so ;;
  esac
}
one() { # lib fmt workload
  sel $1
  fmt=""; [ $1 != base ] && fmt="--visited-format $2"
  timeout 200 python bench.py --only $3 $fmt 2>$O/err_$1_$2_$3.log | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read()); v=list(o.values())[0]
print('$1 fmt$2 $3', {k: (round(v[k],4) if isinstance(v[k], float) else v[k]) for k in v if k in ('avg_kernel_ms','qps','frac_of_hbm_peak','algorithmic_GBps')}, v.get('oracle_sample'))"
  grep "visited cap\|queries per CU" $O/err_$1_$2_$3.log | tail -3
}
for w in u8 sq8; do
  one base 32 $w; one new 32 $w; one new 16 $w; one sgpr96 16 $w; one sgpr80 16 $w; one sgpr80 32 $w
done > $O/ab_int.log 2>&1
cd $R; unset DANN_LIB_PATH
for cfg in "base 0" "new 32" "new 0" "sgpr80 0"; do set -- $cfg
  sel $1
  fmt=""; [ $1 != base ] && fmt="--visited-format $2"
  echo "== $cfg"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --large none $fmt 2>$O/err_main_$1_$2.log | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read())
print({k:o.get(k) for k in ('value','ms_per_step')}, o.get('roofline'))"
  grep "visited cap" $O/err_main_$1_$2.log | tail -2
  timeout 400 python bench.py --only large $fmt 2>$O/err_large_$1_$2.log | tail -1 | python -c "
import sys, json
o=json.loads(sys.stdin.read())['roofline_large']
print({k:o.get(k) for k in ('kernel_ms','avg_kernel_ms','frac','qps','L','recall_at_10','oracle_sample')})"
  grep "visited cap" $O/err_large_$1_$2.log | tail -2
done > $O/ab_f32.log 2>&1

timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 800 python scratch/spill_sweep.py 2>&1 | grep -vE "^$|Warn|amdgpu.ids"
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -3

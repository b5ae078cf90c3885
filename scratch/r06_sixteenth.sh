#!/bin/bash
# round 6, sixteenth call: whole GPU suite at the head (default, then on 16-bit visited tables without teams) + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tee -a $O/pytest.log
DANN_TEST_VISITED_FORMAT=16 DANN_TUNE_OFF=4 timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_ht16.log 2>&1; grep -E "passed|failed" $O/pytest_ht16.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_ht16.log | head

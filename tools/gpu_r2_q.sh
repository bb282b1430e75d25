#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2q; mkdir -p $O
timeout 300 python -m pytest tests/test_deform_pool_gpu.py -q > $O/dp.log 2>&1; echo "deform_pool rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/dp.log | head -20 | tee -a $O/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_deform_pool_gpu.py > $O/suite.log 2>&1; echo "suite rc=$?" | tee -a $O/summary.txt; tail -3 $O/suite.log | tee -a $O/summary.txt
timeout 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; cut -c1-300 $O/bench.json | tee -a $O/summary.txt

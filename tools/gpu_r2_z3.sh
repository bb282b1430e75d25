#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2z3; mkdir -p $O
timeout 600 python -m pytest tests/test_dcn_gpu.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/tests.log | head -20 | tee -a $O/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace0 -- python tools/microbench_dcn.py --iters 5 > $O/mb0.txt 2>&1
grep "all 13\|DCNv2" $O/mb0.txt | tee -a $O/summary.txt
db=$(find $O/trace0 -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/stats0.csv 2>&1; head -6 $O/stats0.csv | cut -c1-140 | tee -a $O/summary.txt; fi
rm -rf $O/trace0

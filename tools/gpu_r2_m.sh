#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2m; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt; tail -15 $O/gpu_suite.log | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -3 $O/smoke.log | tee -a $O/summary.txt

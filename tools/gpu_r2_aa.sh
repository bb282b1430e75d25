#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2aa; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_crnn_gpu.py tests/test_fullsize_parity_gpu.py tests/test_ddp_gpu.py tests/test_dcn_gpu.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/tests.log | head -20 | tee -a $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-260 | tee -a $O/summary.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2u; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "bilstm or lstm" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/tests.log | head -20 | tee -a $O/summary.txt
echo "== xcd map on" | tee -a $O/summary.txt; timeout 200 python tools/microbench_lstm.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
echo "== xcd map off" | tee -a $O/summary.txt; timeout 200 python tools/microbench_lstm.py --no-xcd 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-260 | tee -a $O/summary.txt

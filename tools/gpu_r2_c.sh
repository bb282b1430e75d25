#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
T=3 N=16 timeout 300 python tools/debug_lstm_persist.py > $O/lstm_debug_small.log 2>&1
grep -v amdgpu.ids $O/lstm_debug_small.log | head -40 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q > $O/pipe.log 2>&1; echo "pipeline rc=$?" | tee -a $O/summary.txt
tail -5 $O/pipe.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py -q -s -k "crnn_fp32" > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
grep -E "bias  |max\|d\||margin|decode|passed|failed|Error" $O/fullsize.log | head -40 | tee -a $O/summary.txt

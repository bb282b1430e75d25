#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/debug_lstm_persist.py > $O/lstm_debug.log 2>&1; echo "lstm debug rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/lstm_debug.log | head -40 | tee -a $O/summary.txt
T=3 N=16 timeout 300 python tools/debug_lstm_persist.py > $O/lstm_debug_small.log 2>&1
grep -v amdgpu.ids $O/lstm_debug_small.log | head -20 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_pipeline_gpu.py -q > $O/decode_pipe.log 2>&1; echo "decode/pipeline rc=$?" | tee -a $O/summary.txt
tail -25 $O/decode_pipe.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py -q -s > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
grep -E "^   |max\|d\||margin|decode|drift|passed|failed|Error|error /" $O/fullsize.log | head -150 | tee -a $O/summary.txt

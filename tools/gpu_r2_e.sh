#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2e; mkdir -p $O
T=3 N=16 timeout 300 python tools/debug_lstm_persist.py > $O/lstm_debug_small.log 2>&1
grep -v amdgpu.ids $O/lstm_debug_small.log | grep -v component | head -30 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "bilstm" > $O/lstm_test.log 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt; tail -15 $O/lstm_test.log | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_decode_gpu.py -q > $O/pipe.log 2>&1; echo "pipeline/decode rc=$?" | tee -a $O/summary.txt
tail -5 $O/pipe.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py -q -s -k "crnn" > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
grep -E "^   |max\|d\||margin|decode|drift|passed|failed|Error" $O/fullsize.log | head -120 | tee -a $O/summary.txt

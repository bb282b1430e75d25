#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2d; mkdir -p $O
T=3 N=16 timeout 300 python tools/debug_lstm_persist.py > $O/lstm_debug_small.log 2>&1
grep -v amdgpu.ids $O/lstm_debug_small.log | head -60 | tee -a $O/summary.txt
timeout 300 python tools/debug_pipeline.py > $O/pipe_debug.log 2>&1
grep -v amdgpu.ids $O/pipe_debug.log | head -40 | tee -a $O/summary.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2f; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "bilstm" > $O/lstm_test.log 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/lstm_test.log | tee -a $O/summary.txt
T=33 N=256 timeout 300 python tools/debug_lstm_persist.py 2>&1 | grep -v amdgpu.ids | grep "status\|max|d|" | head -8 | tee -a $O/summary.txt
timeout 300 python tools/microbench_lstm.py 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q > $O/pipe.log 2>&1; echo "pipeline rc=$?" | tee -a $O/summary.txt; tail -3 $O/pipe.log | tee -a $O/summary.txt
timeout 1200 python -m pytest tests/test_fullsize_parity_gpu.py -q -s -k "crnn" > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
grep -E "cnn.0.0.0.weight|max\|d\||margin|decode|drift|passed|failed|Error" $O/fullsize.log | head -20 | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r2f/bench.log").read().strip().splitlines()[-1])
    print("bench ms/step", d["ms_per_step"], "img/s", d["value"])
except Exception as e: print("bench parse failed", e)
PY

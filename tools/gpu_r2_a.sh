#!/bin/bash
# round 2, GPU call A: persistent LSTM correctness + timing, full-size parity tests, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "bilstm" > $O/lstm_test.log 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/microbench_lstm.py > $O/lstm_mb_persist.log 2>&1; tail -4 $O/lstm_mb_persist.log | tee -a $O/summary.txt
timeout 300 python tools/microbench_lstm.py --no-persist > $O/lstm_mb_steps.log 2>&1; tail -4 $O/lstm_mb_steps.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py -q -s > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
grep -E "worst|max\|d\||margin|decode|drift|passed|failed|Error|assert" $O/fullsize.log | head -40 | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r2a/bench.log").read().strip().splitlines()[-1])
    print("bench ms/step", d["ms_per_step"], "img/s", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
    for k,v in d["kernels"].items(): print("  ",k,v)
except Exception as e: print("bench parse failed", e)
PY
timeout 900 python -m pytest tests/test_crnn_gpu.py tests/test_ddp_gpu.py -x -q > $O/crnn_tests.log 2>&1; echo "crnn/ddp tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/crnn_tests.log | tee -a $O/summary.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x > $O/kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -4 $O/kernels.log | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r2h/bench.log").read().strip().splitlines()[-1])
    print("bench ms/step", d["ms_per_step"], "img/s", d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
    for k,v in d["kernels"].items(): print("  ",k,v)
except Exception as e: print("bench parse failed", e)
PY
timeout 900 python -m pytest tests/test_crnn_gpu.py tests/test_fullsize_parity_gpu.py -q -k "crnn" > $O/crnn.log 2>&1; echo "crnn tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/crnn.log | tee -a $O/summary.txt

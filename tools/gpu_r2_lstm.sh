#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_crnn_gpu.py tests/test_fullsize_parity_gpu.py -x -q -k "lstm or crnn or fullsize" 2>&1 | tail -3
run() { echo -n "$1 : "; timeout 200 python tools/bench_with.py $1 -- --no-cpu-baseline --no-secondary --no-kernel-timer 2>&1 | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run "mr_set_tn_taps_group=0"
run "mr_set_tn_taps_group=0"

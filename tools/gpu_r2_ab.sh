#!/bin/bash
run() { echo -n "$1 : "; timeout 200 python tools/bench_with.py $1 -- --no-cpu-baseline --no-secondary --no-kernel-timer 2>&1 | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run "mr_set_tn_taps_group=0"
run "mr_set_nt_p8=1"
run "mr_set_tn_model=0"
run "mr_set_tn_taps_fin=1"
run "mr_set_tn_taps_fin=1 mr_set_tn_taps_group=2"
run "mr_set_lstm_persist=0"

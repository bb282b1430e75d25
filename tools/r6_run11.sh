#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_11; mkdir -p $O
timeout 900 python -m pytest tests/test_persistent_beside_gpu.py -x -q > $O/test_beside.log 2>&1; tail -3 $O/test_beside.log
timeout 300 python bench.py --workload db --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $O/bench_db.json 2>$O/bench_db.err; tail -c 1500 $O/bench_db.json; tail -3 $O/bench_db.err
timeout 300 python bench.py --workload crnn --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/bench_crnn.json 2>$O/bench_crnn.err; python -c "
import json,sys
d=json.loads(open('$O/bench_crnn.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']); print({k:v for k,v in d['kernels'].items() if 'group' in k or 'taps' in k})"
timeout 300 python bench.py --workload fpn_attention --teacher-forcing random --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
MEGREADER_TIMED_STEP_DUMP=$PWD/$O/bf16_drift_timed_step.txt timeout 2400 python -m pytest tests/test_timed_step_gpu.py -x -q -s > $O/test_timed.log 2>&1; tail -5 $O/test_timed.log; grep "group " $O/test_timed.log | head -60

#!/bin/bash
# deterministic tap-split DCN forward: parity, the DB replay-vs-eager difference, DB step time
cd /root/repo
mkdir -p gpurun_out/r5p
O=gpurun_out/r5p
( time timeout 900 python -m pytest tests/test_dcn_gpu.py tests/test_dcn_reference_gpu.py tests/test_deformable_resnet_gpu.py tests/test_seg_detector_gpu.py -x -q 2>&1 | tail -4 ) > $O/pytest1.log 2>&1
tail -5 $O/pytest1.log
( time timeout 900 python -m pytest tests/test_timed_step_gpu.py -x -q -s -k db 2>&1 | grep -v "^   \|amdgpu.ids" | tail -25 ) > $O/pytest_db_timed.log 2>&1
grep -E "replay vs eager|passed|failed|Error|assert|real" $O/pytest_db_timed.log | head
timeout 300 python bench.py --workload db --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-timer 2>$O/bench_db.log | tail -1 > $O/bench_db.json
python -c "import json; d=json.load(open('$O/bench_db.json')); print('db', d['ms_per_step'], d['final_loss'])"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_ddp_gpu.py -x -q -m gpu > $O/pytest_ddp.log 2>&1; tail -3 $O/pytest_ddp.log
MEGREADER_TIMED_STEP_DUMP=$GRAFT_REPO_ROOT/$O/timed_step_dump.txt timeout 1200 python -m pytest tests/test_timed_step_gpu.py -x -q -m gpu -s > $O/pytest_timed.log 2>&1; tail -3 $O/pytest_timed.log
grep "BF16-DRIFT\|loss eager" $O/pytest_timed.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log | cut -c1-300
echo done

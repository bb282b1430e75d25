#!/bin/bash
# full GPU suite + default bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3full; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log | cut -c1-200
timeout 300 python bench.py --workload fpn_attention --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_fpn.log 2>&1; tail -1 $O/bench_fpn.log | cut -c1-200

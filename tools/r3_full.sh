#!/bin/bash
# full GPU suite + the four bench lines (short)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3full; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log | cut -c1-200
for w in crnn res50ppm fpn_attention db; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 40 --warmup 5 > $O/bench_$w.log 2>&1; tail -1 $O/bench_$w.log | cut -c1-230
done

#!/bin/bash
# full GPU suite + default bench with the all-taps wgrad kernel as the default
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_default.json; cut -c1-300 $O/bench_default.json

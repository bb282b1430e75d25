#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dcn_gpu.py tests/test_seg_detector_gpu.py tests/test_dcn_reference_gpu.py -x -q -m gpu \
  -k "conv_transpose or dcn or seg or packed or detector" > $O/pytest1.log 2>&1; tail -5 $O/pytest1.log
timeout 900 python -m pytest tests/test_timed_step_gpu.py tests/test_published_configs_gpu.py -x -q -m gpu -k "db" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
b db "X=1" --workload db --steps 20 --warmup 3
timeout 200 python tools/trace_ops.py --workload db > $O/ops_db.txt 2>&1; head -3 $O/ops_db.txt
echo done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_detector_gpu.py tests/test_dcn_gpu.py -x -q -m gpu > $O/pytest1.log 2>&1; tail -15 $O/pytest1.log
timeout 900 python -m pytest tests/test_timed_step_gpu.py tests/test_published_configs_gpu.py -x -q -m gpu -k "db" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
b db "X=1" --workload db --steps 20 --warmup 3
b db_nofuse "MEGREADER_DB_LOSS_FUSED=0 MEGREADER_DB_TAIL_FUSED=0" --workload db --steps 20 --warmup 3
echo done

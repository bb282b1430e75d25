#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_crnn_gpu.py tests/test_kernels_gpu.py tests/test_stem_gpu.py tests/test_dropin_fast_gpu.py -x -q -m gpu \
  -k "crnn or adam or optim or sgd or prep or trajectory or dropin" > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
timeout 600 python -m pytest tests/test_timed_step_gpu.py -x -q -m gpu -k "crnn or db" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
b crnn "X=1" --no-secondary --steps 40 --warmup 5
b crnn_b32 "X=1" --no-secondary --steps 40 --warmup 5 --batch 32
for w in res50ppm fpn_attention db; do b $w "X=1" --workload $w --steps 15 --warmup 3; done
echo done

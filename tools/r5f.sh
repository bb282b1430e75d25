#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_crnn_gpu.py -x -q -m gpu -k "conv or crnn or gemm or nt_" > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
for w8 in 3 7 11 19 35 63; do
b crnn_w8_$w8 "MEGREADER_TUNING=nt_wide8=$w8" --no-secondary --steps 40 --warmup 5
b crnn_b32_w8_$w8 "MEGREADER_TUNING=nt_wide8=$w8" --no-secondary --steps 40 --warmup 5 --batch 32
for w in res50ppm fpn_attention db; do
b ${w}_w8_$w8 "MEGREADER_TUNING=nt_wide8=$w8" --workload $w --steps 15 --warmup 3
done; done
echo done

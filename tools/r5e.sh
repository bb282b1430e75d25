#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_tn_grouped_gpu.py tests/test_crnn_gpu.py tests/test_kernels_gpu.py -x -q -m gpu \
  -k "grouped or crnn or ctc or linear or bilstm or defer or gemm_tn2 or conv" > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
for w8 in 0 1 2 3; do
b crnn_w8_$w8 "MEGREADER_TUNING=nt_wide8=$w8" --no-secondary --steps 40 --warmup 5
done
for w in res50ppm fpn_attention db; do for w8 in 0 1 2 3; do
b ${w}_w8_$w8 "MEGREADER_TUNING=nt_wide8=$w8" --workload $w --steps 15 --warmup 3
done; done
b crnn_b32_w8_0 "MEGREADER_TUNING=nt_wide8=0" --no-secondary --steps 40 --warmup 5 --batch 32
b crnn_b32_w8_2 "MEGREADER_TUNING=nt_wide8=2" --no-secondary --steps 40 --warmup 5 --batch 32
echo done

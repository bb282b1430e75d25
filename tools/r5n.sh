#!/bin/bash
# pool bwd kernel, _SeqLinear on the prepared-image cache, side-stream flush of deferred weight gradients
cd /root/repo
mkdir -p gpurun_out/r5n
O=gpurun_out/r5n
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_attention_kernels_gpu.py tests/test_fpn_attention_gpu.py tests/test_tn_grouped_gpu.py tests/test_crnn_gpu.py tests/test_timed_step_gpu.py -x -q 2>&1 | tail -5 > $O/pytest1.log
tail -3 $O/pytest1.log
b() { # name env... -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-timer "$@" 2>$O/$name.log | tail -1 > $O/$name.json
  python -c "import sys,json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['final_loss'])" 2>/dev/null || { echo "$name FAILED"; tail -5 $O/$name.log; }
}
for s in 0 1 2; do
  b crnn_side$s MEGREADER_TN_SIDE=$s -- --workload crnn
  b crnn_b32_side$s MEGREADER_TN_SIDE=$s -- --workload crnn --batch 32
done
b crnn_b32_side2_m2 MEGREADER_TN_SIDE=2 MEGREADER_TN_SIDE_MAX=2 -- --workload crnn --batch 32
b crnn_b32_side2_m4 MEGREADER_TN_SIDE=2 MEGREADER_TN_SIDE_MAX=4 -- --workload crnn --batch 32
for w in res50ppm fpn_attention db; do
  for s in 1 2; do
    b ${w}_side$s MEGREADER_TN_SIDE=$s -- --workload $w
  done
done
b res50ppm_side2_m6 MEGREADER_TN_SIDE=2 MEGREADER_TN_SIDE_MAX=6 -- --workload res50ppm

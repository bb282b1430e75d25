#!/bin/bash
# FPN-attention: output layer of the teacher-forced decode loop as one GEMM + one log-softmax/NLL launch behind the loop
cd /root/repo
mkdir -p gpurun_out/r5u
O=gpurun_out/r5u
b() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-timer "$@" 2>$O/$name.log | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['final_loss'])" 2>/dev/null || { echo "$name FAILED"; tail -5 $O/$name.log; }; }
b fpn_batched0 MEGREADER_DECODE_BATCHED_OUT=0 -- --workload fpn_attention
b fpn_batched1 MEGREADER_DECODE_BATCHED_OUT=1 -- --workload fpn_attention
timeout 900 python -m pytest tests/test_fpn_attention_gpu.py tests/test_attention_kernels_gpu.py tests/test_published_configs_gpu.py tests/test_timed_step_gpu.py -x -q -k "fpn or attention or attn" 2>&1 | tail -3

#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_tn_taps_gpu.py -x -q -k "group_reduction or gemm_tn or wgrad or taps" > gpurun_out/tn_group_test.log 2>&1; tail -3 gpurun_out/tn_group_test.log
timeout 200 python tools/microbench_conv.py --layers 1,6 --only wgradtab 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | tail -1 > gpurun_out/bench_tn_group.json; python -c "
import json; d=json.load(open('gpurun_out/bench_tn_group.json')); print(d['value'], d['ms_per_step'], d['secondary']['value'], d['secondary']['ms_per_step'])
for k,v in d['secondary']['kernels'].items():
    if 'tn' in k: print(k, v)"

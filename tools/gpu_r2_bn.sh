#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "batchnorm" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_res50ppm_gpu.py tests/test_crnn_gpu.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep "^{" | tail -1 > gpurun_out/bench_bn_fused.json; python -c "
import json; d=json.load(open('gpurun_out/bench_bn_fused.json')); print(d['value'], d['ms_per_step'], d['secondary']['value'], d['secondary']['ms_per_step'])"

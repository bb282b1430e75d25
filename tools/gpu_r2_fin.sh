#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tn_taps_gpu.py -x -q > gpurun_out/taps_test.log 2>&1; tail -3 gpurun_out/taps_test.log
for f in 0 1; do for g in 2 4 8; do
  echo "== fin $f group $g"
  timeout 300 python tools/microbench_tn_taps.py --fin $f --group $g --sweep 0 --layers crnn.conv2,crnn.conv3,crnn.conv4,crnn.conv5 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/taps_bench_fin.log

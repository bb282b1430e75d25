#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tn_taps_gpu.py -x -q > gpurun_out/taps_test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/taps_test.log
tail -8 gpurun_out/taps_test.log
for g in 2 4 8; do
  echo "== w8 group $g"
  timeout 300 python tools/microbench_tn_taps.py --w8 1 --group $g --layers crnn.conv2,crnn.conv3,crnn.conv4,crnn.conv5 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/taps_bench_w8.log

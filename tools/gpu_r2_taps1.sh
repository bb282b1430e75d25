#!/bin/bash
# all-taps wgrad kernel: correctness first, then the microbench
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tn_taps_gpu.py -x -q > gpurun_out/taps_test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/taps_test.log
tail -15 gpurun_out/taps_test.log
timeout 300 python tools/microbench_tn_taps.py > gpurun_out/taps_bench.log 2>&1
cat gpurun_out/taps_bench.log

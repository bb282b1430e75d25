#!/bin/bash
# float32 NT kernels with a float64 running total: per-layer error, then the f32 parity tests
cd /root/repo
mkdir -p gpurun_out/r5o
O=gpurun_out/r5o
python tools/diag_f32_error.py 2>&1 | grep -v amdgpu.ids | tee $O/diag_f32_error.txt
( time timeout 1200 python -m pytest tests/test_published_configs_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -60 ) > $O/pytest_published.log 2>&1
grep -E "passed|failed|error|layer|PPM|stage|real" $O/pytest_published.log | head -40
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_crnn_gpu.py tests/test_res50ppm_gpu.py -x -q 2>&1 | tail -5 ) > $O/pytest2.log 2>&1
tail -6 $O/pytest2.log

#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "big_tile or head_tail" 2>&1 | tail -3
timeout 200 python tools/microbench_conv.py --layers 2,4 --only fwd,dgrad 2>&1 | grep -v amdgpu.ids

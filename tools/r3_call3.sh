#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c3; mkdir -p $O
timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py tests/test_seg_detector_gpu.py -m gpu -q -s > $O/pytest_dcn.log 2>&1
tail -12 $O/pytest_dcn.log
timeout 200 python tools/microbench_dcn.py --batch 16 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_fused_b16.txt; cat $O/dcn_microbench_fused_b16.txt | cut -c1-250
timeout 200 python tools/microbench_dcn.py --batch 2 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_fused_b2.txt; tail -1 $O/dcn_microbench_fused_b2.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_dcn -- python tools/microbench_dcn.py --batch 16 --iters 5 > $O/trace_dcn.log 2>&1
db=$(find $O/trace_dcn -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/dcn_microbench_kernel_stats.csv 2>&1; head -12 $O/dcn_microbench_kernel_stats.csv | cut -c1-150; fi
rm -rf $O/trace_dcn

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c6; mkdir -p $O
timeout 300 python -m pytest tests/test_deformable_resnet_gpu.py -m gpu -q > $O/pytest_def1.log 2>&1; tail -3 $O/pytest_def1.log
timeout 300 python -m pytest tests/test_deformable_resnet_gpu.py -m gpu -q > $O/pytest_def2.log 2>&1; tail -3 $O/pytest_def2.log
timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py tests/test_seg_detector_gpu.py -m gpu -q > $O/pytest_dcn.log 2>&1; tail -8 $O/pytest_dcn.log
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_dcn -- python tools/microbench_dcn.py --batch 16 --iters 5 > $O/trace_dcn.log 2>&1
grep "layer\|all 13" $O/trace_dcn.log | cut -c1-60,150-260
db=$(find $O/trace_dcn -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/dcn_microbench_kernel_stats.csv 2>&1; head -8 $O/dcn_microbench_kernel_stats.csv | cut -c1-150; fi
rm -rf $O/trace_dcn

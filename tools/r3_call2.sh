#!/bin/bash
# Round-3 GPU call 2: fused DCN kernels (parity + microbench + kernel stats), published-config parity tests, DB workload as a hipGraph.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c2; mkdir -p $O
timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py tests/test_seg_detector_gpu.py -m gpu -q -x -s > $O/pytest_dcn.log 2>&1
tail -15 $O/pytest_dcn.log
timeout 200 python tools/microbench_dcn.py --batch 16 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_fused_b16.txt; cat $O/dcn_microbench_fused_b16.txt
timeout 200 python tools/microbench_dcn.py --batch 2 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_fused_b2.txt; tail -1 $O/dcn_microbench_fused_b2.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_dcn -- python tools/microbench_dcn.py --batch 16 --iters 5 > $O/trace_dcn.log 2>&1
db=$(find $O/trace_dcn -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/dcn_microbench_kernel_stats.csv 2>&1; head -16 $O/dcn_microbench_kernel_stats.csv | cut -c1-150; fi
rm -rf $O/trace_dcn
timeout 600 python -m pytest tests/test_published_configs_gpu.py -m gpu -q -s > $O/pytest_pub.log 2>&1
grep -n "fp32\|worst\|passed\|failed\|Error" $O/pytest_pub.log | head -40
timeout 300 python bench.py --workload db --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_db.log 2>&1; tail -1 $O/bench_db.log | cut -c1-400

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c9; mkdir -p $O
timeout 600 python -m pytest tests/test_attention_kernels_gpu.py tests/test_fpn_attention_gpu.py "tests/test_published_configs_gpu.py::test_fpn_attention_fp32_n32_elementwise" -m gpu -q -s > $O/pytest_attn.log 2>&1
grep -E "passed|failed|Error|error|FPN50|worst" $O/pytest_attn.log | cut -c1-300 | head -30
timeout 300 python bench.py --workload fpn_attention --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_fpn.log 2>&1; tail -1 $O/bench_fpn.log | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fpn -- python bench.py --workload fpn_attention --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/trace_fpn.log 2>&1
db=$(find $O/trace_fpn -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/fpn_attention_kernel_stats.csv 2>&1; head -30 $O/fpn_attention_kernel_stats.csv | cut -c1-160; tail -1 $O/fpn_attention_kernel_stats.csv; fi
rm -rf $O/trace_fpn

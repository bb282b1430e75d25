#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c11; mkdir -p $O
timeout 600 python -m pytest tests/test_dropin_fast_gpu.py tests/test_attention_kernels_gpu.py tests/test_fpn_attention_gpu.py -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|^E " $O/pytest.log | cut -c1-300 | head -20
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fpn -- python bench.py --workload fpn_attention --no-cpu-baseline --no-kernel-timer --steps 10 --warmup 3 > $O/trace_fpn.log 2>&1
tail -1 $O/trace_fpn.log | cut -c1-230
db=$(find $O/trace_fpn -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/fpn_attention_kernel_stats.csv 2>&1; head -24 $O/fpn_attention_kernel_stats.csv | cut -c1-160; tail -1 $O/fpn_attention_kernel_stats.csv; fi
rm -rf $O/trace_fpn

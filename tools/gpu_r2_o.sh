#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_detector_gpu.py tests/test_dcn_gpu.py tests/test_fpn_attention_gpu.py -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed" $O/tests.log | head -20 | tee -a $O/summary.txt
timeout 900 python bench.py --workload db --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_db.log 2>$O/bench_db.err; echo "db bench rc=$?" | tee -a $O/summary.txt; tail -1 $O/bench_db.log | cut -c1-900 | tee -a $O/summary.txt; tail -4 $O/bench_db.err | cut -c1-300 | tee -a $O/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --workload db --no-cpu-baseline --no-kernel-timer --steps 4 --warmup 2 > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" 6 > $O/db_kernel_stats.csv 2>&1; head -30 $O/db_kernel_stats.csv | cut -c1-180 | tee -a $O/summary.txt; fi
rm -rf $O/trace

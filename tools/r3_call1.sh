#!/bin/bash
# Round-3 GPU call 1: the new parity tests at the published configurations + kernel-level attention tests + DCN at the real
# layer shapes (existing kernels), and rocprofv3 kernel stats of the fpn_attention / db workloads.  Outputs: gpurun_out/r3c1/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c1; mkdir -p $O
nproc > $O/host.txt; free -g >> $O/host.txt
timeout 900 python -m pytest tests/test_attention_kernels_gpu.py tests/test_ctc2d_gpu.py tests/test_dcn_gpu.py tests/test_published_configs_gpu.py -m gpu -q -s --durations=15 > $O/pytest_new.log 2>&1
tail -40 $O/pytest_new.log
for w in fpn_attention db; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -- python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/trace_$w.log 2>&1
  db=$(find $O/trace_$w -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" 13 > $O/${w}_kernel_stats.csv 2>&1; head -25 $O/${w}_kernel_stats.csv | cut -c1-170; fi
  tail -1 $O/trace_$w.log | cut -c1-300
  rm -rf $O/trace_$w
done

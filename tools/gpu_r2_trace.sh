#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02t; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/crnn_kernel_stats.csv 2>&1; head -60 $O/crnn_kernel_stats.csv | cut -c1-170; fi
rm -rf $O/trace

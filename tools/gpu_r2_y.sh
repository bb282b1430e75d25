#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2y; mkdir -p $O
timeout 300 python tools/microbench_dcn.py 2>&1 | grep -v amdgpu.ids | tee $O/dcn_microbench.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python tools/microbench_dcn.py --iters 5 > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/dcn_kernel_stats.csv 2>&1; head -14 $O/dcn_kernel_stats.csv | cut -c1-150; fi
rm -rf $O/trace

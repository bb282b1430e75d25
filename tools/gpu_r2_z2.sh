#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2z2; mkdir -p $O
for v in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$v -- python tools/microbench_dcn.py --iters 5 --v1 $v > $O/mb$v.txt 2>&1
  grep "all 13\|DCNv2" $O/mb$v.txt | tee -a $O/summary.txt
  db=$(find $O/trace$v -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/stats$v.csv 2>&1; head -8 $O/stats$v.csv | cut -c1-140 | tee -a $O/summary.txt; fi
  rm -rf $O/trace$v
done

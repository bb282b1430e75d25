#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
for attempt in 1 2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_db -- python bench.py --workload db --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace_db.log 2>&1
  db=$(find $O/trace_db -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/db_kernel_stats.csv 2>&1; head -3 $O/db_kernel_stats.csv | cut -c1-150; tail -1 $O/db_kernel_stats.csv; rm -rf $O/trace_db; break; fi
  echo "attempt $attempt failed: $(grep -c SIGSEGV $O/trace_db.log) segv"; rm -rf $O/trace_db
done
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_default.json; cut -c1-200 $O/bench_default.json; grep -o '"traffic": [^,]*' $O/bench_default.json

#!/bin/bash
# Round-2 measurement: rocprofv3 kernel trace of the default CRNN bench command, FETCH_SIZE / WRITE_SIZE PMC passes
# (separate passes, no trace domains combined with --pmc) for both north-star workloads.  Outputs: gpurun_out/r02p/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02p; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --no-cpu-baseline --no-secondary > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/crnn_kernel_stats.csv 2>&1; head -8 $O/crnn_kernel_stats.csv | cut -c1-160; fi
rm -rf $O/trace
for w in crnn res50ppm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python bench.py --workload $w --no-secondary --no-cpu-baseline --no-graph --no-kernel-timer --steps 3 --warmup 2 > $O/pmc_${w}_$c.log 2>&1
    f=$(find $O/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc_${w}_$c.txt 2>&1; fi
    rm -rf $O/pmc_${w}_$c
  done
  python tools/pmc_to_json.py $O/pmc_${w}_FETCH_SIZE.txt $O/pmc_${w}_WRITE_SIZE.txt $O/pmc_traffic_${w}.json > /dev/null 2>&1
  grep -A3 "igemm" $O/pmc_${w}_FETCH_SIZE.txt | head -30
done

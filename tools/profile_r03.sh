#!/bin/bash
# Round-3 measurement (GPU box): default bench line, rocprofv3 kernel traces of the four workloads, FETCH_SIZE / WRITE_SIZE
# PMC passes of the two north-star workloads (separate passes; --pmc is never combined with a trace domain), DCN microbench.
# Outputs: gpurun_out/r03p/ (copy what should be judged into profiles/).   usage: bash tools/profile_r03.sh [quick]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03p; mkdir -p $O
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_default.json; cut -c1-260 $O/bench_default.json
for w in crnn res50ppm fpn_attention db; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -- python bench.py --workload $w --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace_$w.log 2>&1
  db=$(find $O/trace_$w -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/${w}_kernel_stats.csv 2>&1; head -4 $O/${w}_kernel_stats.csv | cut -c1-150; tail -1 $O/${w}_kernel_stats.csv; fi
  grep -o '"ms_per_step": [0-9.]*' $O/trace_$w.log | head -1
  rm -rf $O/trace_$w
done
if [ "$1" != "quick" ]; then
for w in crnn res50ppm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python bench.py --workload $w --no-secondary --no-cpu-baseline --no-graph --no-kernel-timer --steps 3 --warmup 2 > $O/pmc_${w}_$c.log 2>&1
    f=$(find $O/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc_${w}_$c.txt 2>&1; fi
    rm -rf $O/pmc_${w}_$c
  done
  python tools/pmc_to_json.py $O/pmc_${w}_FETCH_SIZE.txt $O/pmc_${w}_WRITE_SIZE.txt $O/pmc_traffic_${w}.json > /dev/null 2>&1
done
fi
for w in fpn_attention db; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$w.log 2>&1; tail -1 $O/bench_$w.log > $O/bench_$w.json; cut -c1-200 $O/bench_$w.json
done
timeout 300 python tools/microbench_dcn.py --batch 16 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_b16.txt; tail -1 $O/dcn_microbench_b16.txt
timeout 300 python tools/microbench_dcn.py --batch 2 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_b2.txt; tail -1 $O/dcn_microbench_b2.txt

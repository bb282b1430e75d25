#!/bin/bash
# Round-4 measurement (GPU box).  Outputs: gpurun_out/r04p/ (copy what should be judged into profiles/).
#   default bench line (CRNN + the three secondaries, CPU baselines) -> bench_default.json
#   rocprofv3 kernel traces (--kernel-trace --stats) of the four workloads at the benchmarked batch, and of the CRNN at per-GPU
#     batch 128 / 64 / 32 (what each rank runs under --scaling strong on 2 / 4 / 8 GPUs: reference data/data_loader.py:40-48)
#   FETCH_SIZE / WRITE_SIZE PMC passes of the two north-star workloads (separate passes; --pmc is never combined with a trace
#     domain) -> pmc_traffic_<workload>.json stamped with the kernel-source hash (tools/pmc_to_json.py)
# usage: bash tools/profile_r04.sh [quick]      (quick: no PMC passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_default.json; cut -c1-200 $O/bench_default.json
trace() {   # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$name -- python bench.py "$@" --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace_$name.log 2>&1
  local db=$(find $O/trace_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/${name}_kernel_stats.csv 2>&1; head -3 $O/${name}_kernel_stats.csv | cut -c1-150; tail -1 $O/${name}_kernel_stats.csv; fi
  grep -o '"ms_per_step": [0-9.]*' $O/trace_$name.log | head -1
  rm -rf $O/trace_$name
}
for w in crnn res50ppm fpn_attention db; do trace $w --workload $w; done
for b in 128 64 32; do
  trace crnn_b$b --workload crnn --batch $b
  timeout 200 python bench.py --workload crnn --batch $b --no-cpu-baseline --no-secondary --steps 40 --warmup 5 > $O/bench_crnn_b$b.log 2>&1; tail -1 $O/bench_crnn_b$b.log > $O/bench_crnn_b$b.json; cut -c1-160 $O/bench_crnn_b$b.json
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
  ls -la $O/pmc_traffic_${w}.json
done
fi

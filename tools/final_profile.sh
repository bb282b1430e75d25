#!/bin/bash
# Round-end measurement: GPU tests, the default bench line, a rocprofv3 kernel trace of the same command and two
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes as gfx950 requires).  Outputs under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
mkdir -p $out
timeout 500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json; cut -c1-400 $out/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -- python bench.py --no-cpu-baseline > $out/trace.log 2>&1
db=$(find $out/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $out/kernel_stats.csv 2>&1; head -12 $out/kernel_stats.csv | cut -c1-150; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python bench.py --no-cpu-baseline --no-graph --no-kernel-timer --steps 3 --warmup 2 > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $out/pmc_$c.txt 2>&1; grep -A2 "igemm\|lstm_step" $out/pmc_$c.txt | head -40; fi
  rm -rf $out/pmc_$c
done
python tools/pmc_to_json.py $out/pmc_FETCH_SIZE.txt $out/pmc_WRITE_SIZE.txt $out/pmc_traffic.json > /dev/null 2>&1
rm -rf $out/trace

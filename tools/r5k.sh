#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_seg_detector_gpu.py -x -q -m gpu -k "conv_transpose or seg" > $O/pytest1.log 2>&1; tail -2 $O/pytest1.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
b db "X=1" --workload db --steps 20 --warmup 3
b db2 "X=1" --workload db --steps 20 --warmup 3
timeout 300 rocprofv3 --kernel-trace -d $O/trace_db -- python bench.py --workload db --no-cpu-baseline --no-secondary --no-kernel-timer --steps 6 --warmup 2 > $O/trace_db.log 2>&1
db=$(find $O/trace_db -name "*.db" | head -1)
python tools/rocpd_sequence.py "$db" --marker sgd_kernel > $O/db_step_sequence.txt 2>&1
python tools/rocpd_stats.py "$db" 8 > $O/db_kernel_stats.csv 2>&1
rm -rf $O/trace_db
head -2 $O/db_step_sequence.txt
echo done

#!/bin/bash
# Round-5 GPU call A: parity of the grouped weight-gradient launches / pool kernels / CTC glue, CRNN A/B, glue traces.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_tn_grouped_gpu.py tests/test_crnn_gpu.py "tests/test_kernels_gpu.py" -x -q -m gpu \
  -k "grouped or crnn or maxpool or ctc or linear or bilstm or defer" > $O/pytest1.log 2>&1; tail -5 $O/pytest1.log
timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_timed_step_gpu.py -x -q -m gpu -k "crnn" > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1)"
}
b crnn_default "X=1" --no-secondary --steps 40 --warmup 5
b crnn_nodefer "MEGREADER_TUNING=tn_defer=0" --no-secondary --steps 40 --warmup 5
b crnn_oldpool "MEGREADER_TUNING=pool_fixed=0" --no-secondary --steps 40 --warmup 5
b crnn_b32 "X=1" --no-secondary --steps 40 --warmup 5 --batch 32
for w in res50ppm fpn_attention db; do
  b ${w}_default "X=1" --workload $w --steps 15 --warmup 3
  b ${w}_nodefer "MEGREADER_TUNING=tn_defer=0" --workload $w --steps 15 --warmup 3
done
for w in crnn db res50ppm; do timeout 200 python tools/trace_glue.py --workload $w --top 80 > $O/glue_$w.txt 2>&1; done
trace() {   # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$name -- python bench.py "$@" --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace_$name.log 2>&1
  local db=$(find $O/trace_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/${name}_kernel_stats.csv 2>&1; fi
  rm -rf $O/trace_$name
}
trace crnn --workload crnn
trace res50ppm --workload res50ppm
echo done

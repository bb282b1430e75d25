#!/bin/bash
# Round-5 GPU call B: CTC linear-domain parity, grouped LSTM test, op traces (forward + backward), CRNN bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_tn_grouped_gpu.py tests/test_crnn_gpu.py tests/test_ctc_decoder_gpu.py tests/test_kernels_gpu.py -x -q -m gpu \
  -k "grouped or crnn or ctc or linear or bilstm or defer or decoder" > $O/pytest1.log 2>&1; tail -5 $O/pytest1.log
b() { # name, env, args
  local name=$1; local envs=$2; shift; shift
  env $envs timeout 300 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.log 2>&1
  tail -1 $O/bench_$name.log > $O/bench_$name.json
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$name.json | head -1) $(grep -o '"final_loss": [0-9.]*' $O/bench_$name.json | head -1)"
}
b crnn_default "X=1" --no-secondary --steps 40 --warmup 5
b crnn_logctc "MEGREADER_TUNING=ctc_linear=0" --no-secondary --steps 40 --warmup 5
b crnn_nodefer "MEGREADER_TUNING=tn_defer=0" --no-secondary --steps 40 --warmup 5
b crnn_b32 "X=1" --no-secondary --steps 40 --warmup 5 --batch 32
for w in crnn db fpn_attention res50ppm; do timeout 200 python tools/trace_ops.py --workload $w > $O/ops_$w.txt 2>&1; done
echo done

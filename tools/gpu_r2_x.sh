#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2x; mkdir -p $O
for m in 0 1; do
  timeout 300 python bench.py --workload res50ppm --no-cpu-baseline --tn-model $m --steps 10 --warmup 3 > $O/b$m.log 2>&1
  echo "res50 tn_model=$m: $(tail -1 $O/b$m.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels']['igemm_tn_kernel<bf16,conv>'])")" | tee -a $O/summary.txt
done
timeout 300 python bench.py --workload fpn_attention --no-cpu-baseline --tn-model 1 --steps 10 --warmup 3 > $O/attn1.log 2>&1; echo "attn tn_model=1: $(tail -1 $O/attn1.log | cut -c1-200)" | tee -a $O/summary.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3probe; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_res50ppm_gpu.py tests/test_dropin_fast_gpu.py tests/test_fpn_attention_gpu.py tests/test_attention_kernels_gpu.py tests/test_crnn_gpu.py -m gpu -q -s > $O/pytest.log 2>&1; tail -4 $O/pytest.log; grep "drop-in train_step" $O/pytest.log
for w in res50ppm fpn_attention; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-kernel-timer --steps 40 --warmup 5 > $O/b_$w.log 2>&1
  echo "$w $(tail -1 $O/b_$w.log | grep -o '"ms_per_step": [0-9.]*')"
done

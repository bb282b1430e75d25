#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_12; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_pool_gpu.py tests/test_crnn_gpu.py -x -q > $O/test_pool.log 2>&1; tail -5 $O/test_pool.log
B="--no-cpu-baseline --no-secondary --no-kernel-timer --steps 30 --warmup 5"
for env in 1 0 1 0; do
  ms=$(MEGREADER_CONV_POOL=$env timeout 300 python bench.py --workload crnn $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "MEGREADER_CONV_POOL=$env crnn $ms"
done > $O/ab_pool.txt 2>&1; cat $O/ab_pool.txt
ms=$(MEGREADER_CONV_POOL=1 timeout 300 python bench.py --workload crnn --batch 32 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "pool=1 crnn b32 $ms"
ms=$(MEGREADER_CONV_POOL=0 timeout 300 python bench.py --workload crnn --batch 32 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "pool=0 crnn b32 $ms"

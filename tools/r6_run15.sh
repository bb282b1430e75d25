#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_15; mkdir -p $O
timeout 1200 python -m pytest tests/test_dcn_gcol_gpu.py tests/test_dcn_gpu.py tests/test_dcn_reference_gpu.py tests/test_deformable_resnet_gpu.py -x -q > $O/test_dcn.log 2>&1; tail -3 $O/test_dcn.log
B="--no-cpu-baseline --no-secondary --no-kernel-timer --steps 30 --warmup 5"
for i in 1 2; do ms=$(timeout 300 python bench.py --workload db $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "db $ms"; done
timeout 600 python tools/microbench_dcn.py --batch 2 2>/dev/null | tail -1

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_17; mkdir -p $O
timeout 1200 python -m pytest tests/test_attention_kernels_gpu.py tests/test_fpn_attention_gpu.py -x -q > $O/test_attn.log 2>&1; tail -3 $O/test_attn.log
B="--no-cpu-baseline --no-secondary --no-kernel-timer --steps 30 --warmup 5"
for i in 1 2 3; do ms=$(timeout 300 python bench.py --workload fpn_attention $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "fpn $ms"; done
ms=$(timeout 300 python bench.py --workload fpn_attention --teacher-forcing random $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "fpn random coins $ms"
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python bench.py --workload fpn_attention --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > $O/fpn_kernel_stats.csv 2>&1; rm -rf $O/trace; grep -n "attn_\|skinny" $O/fpn_kernel_stats.csv | cut -c1-160

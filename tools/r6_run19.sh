#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_19; mkdir -p $O
timeout 900 python -m pytest tests/test_fpn_attention_gpu.py -x -q > $O/test.log 2>&1; tail -2 $O/test.log
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python bench.py --workload fpn_attention --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1); python tools/rocpd_sequence.py "$db" --marker adam_kernel > $O/fpn_seq.txt 2>&1; rm -rf $O/trace; head -1 $O/fpn_seq.txt
awk '{ if ($2=="gap" && $3+0>3) print }' $O/fpn_seq.txt | cut -c1-130
for i in 1 2; do timeout 300 python bench.py --workload fpn_attention --no-cpu-baseline --no-secondary --no-kernel-timer --steps 30 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done

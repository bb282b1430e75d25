#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
for b in 256 32; do
timeout 300 rocprofv3 --kernel-trace -d $O/trace_crnn$b -- python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --steps 6 --warmup 2 --batch $b > $O/trace_crnn$b.log 2>&1
db=$(find $O/trace_crnn$b -name "*.db" | head -1)
python tools/rocpd_sequence.py "$db" > $O/crnn_b${b}_step_sequence.txt 2>&1
python tools/rocpd_stats.py "$db" 8 > $O/crnn_b${b}_kernel_stats.csv 2>&1
rm -rf $O/trace_crnn$b
done
echo done

#!/bin/bash
# refresh the FPN-attention trace after the output layer left the decode recurrence
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05p; mkdir -p $O
name=fpn_attention
timeout 300 rocprofv3 --kernel-trace -d $O/trace_$name -- python bench.py --workload fpn_attention --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace_$name.log 2>&1
db=$(find $O/trace_$name -name "*.db" | head -1)
python tools/rocpd_stats.py "$db" > $O/${name}_kernel_stats.csv 2>&1
python tools/rocpd_sequence.py "$db" --marker adam_kernel > $O/${name}_step_sequence.txt 2>&1
head -1 $O/${name}_step_sequence.txt
grep -o '"ms_per_step": [0-9.]*' $O/trace_$name.log | head -1
rm -rf $O/trace_$name

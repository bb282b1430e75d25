#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_13; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_pool_gpu.py -x -q -rs > $O/test_pool.log 2>&1; tail -8 $O/test_pool.log
timeout 300 rocprofv3 --kernel-trace -d $O/trace_crnn -- python bench.py --workload crnn --no-cpu-baseline --no-secondary --no-kernel-timer --steps 10 --warmup 3 > $O/trace.log 2>&1
db=$(find $O/trace_crnn -name "*.db" | head -1); python tools/rocpd_sequence.py "$db" --marker adam_kernel > $O/crnn_step_sequence.txt 2>&1; rm -rf $O/trace_crnn; head -24 $O/crnn_step_sequence.txt | cut -c1-150

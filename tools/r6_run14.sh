#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_14; mkdir -p $O
timeout 1200 python -m pytest tests/test_dcn_gcol_gpu.py tests/test_dcn_gpu.py tests/test_dcn_reference_gpu.py -x -q > $O/test_dcn.log 2>&1; tail -4 $O/test_dcn.log
timeout 600 python tools/microbench_dcn.py --batch 16 > $O/dcn_b16.txt 2>&1; tail -7 $O/dcn_b16.txt
timeout 600 python tools/microbench_dcn.py --batch 2 > $O/dcn_b2.txt 2>&1; tail -1 $O/dcn_b2.txt
B="--no-cpu-baseline --no-secondary --no-kernel-timer --steps 30 --warmup 5"
for cfg in "dcn_col_fwd=1" "dcn_col_fwd=0" "dcn_gcol=0" "dcn_col_fwd=1"; do
  ms=$(MEGREADER_TUNING=$cfg timeout 300 python bench.py --workload db $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); echo "$cfg db $ms"
done > $O/ab_db.txt 2>&1; cat $O/ab_db.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace -d $O/trace_dcn -- python tools/microbench_dcn.py --batch 16 --iters 3 > $O/trace_dcn.log 2>&1
db=$(find $O/trace_dcn -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" > $O/dcn_b16_kernel_stats.csv 2>&1; rm -rf $O/trace_dcn; head -14 $O/dcn_b16_kernel_stats.csv | cut -c1-150

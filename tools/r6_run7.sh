#!/bin/bash
# round 6, call 7: fixed nt32 test; in-step A/B of the all-taps reduction variants and of nt_m32; DCN microbench baseline
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_7; mkdir -p $O
timeout 900 python -m pytest tests/test_nt32_gpu.py -x -q > $O/test_nt32.log 2>&1; tail -2 $O/test_nt32.log
B="--no-cpu-baseline --no-secondary --no-kernel-timer --steps 30 --warmup 5"
for cfg in "nt_m32=0" "nt_m32=1" "tn_taps_w8=1" "tn_taps_fin=2" "tn_taps_fin=1" "nt_m32=0"; do
  for wl in crnn res50ppm; do
    ms=$(MEGREADER_TUNING=$cfg timeout 300 python bench.py --workload $wl $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$cfg $wl $ms"
  done
done > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
timeout 600 python tools/microbench_dcn.py --batch 16 > $O/dcn_b16.txt 2>&1; tail -8 $O/dcn_b16.txt

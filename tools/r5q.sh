#!/bin/bash
# BN fused apply passes: rows per workgroup / block cap sweep in the Res50-PPM step; DCN tests after the slab change
cd /root/repo
mkdir -p gpurun_out/r5q
O=gpurun_out/r5q
timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_dcn_reference_gpu.py -x -q 2>&1 | tail -3
b() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-timer "$@" 2>$O/$name.log | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'])" 2>/dev/null || { echo "$name FAILED"; tail -3 $O/$name.log; }; }
for g in 2 4 8 16 32; do b res50ppm_g$g MEGREADER_BN_GROUPS=$g -- --workload res50ppm; done
b res50ppm_g4_cap8k MEGREADER_BN_GROUPS=4 MEGREADER_BN_CAP=8192 -- --workload res50ppm
b res50ppm_g8_cap2k MEGREADER_BN_GROUPS=8 MEGREADER_BN_CAP=2048 -- --workload res50ppm
b res50ppm_g16_cap1k MEGREADER_BN_GROUPS=16 MEGREADER_BN_CAP=1024 -- --workload res50ppm

#!/bin/bash
# split reduction also instead of the 8-wave variants for few-tile launches: parity + A/B
cd /root/repo
mkdir -p gpurun_out/r5t
O=gpurun_out/r5t
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "split_reduction or gemm_nt or addend or conv_relu" 2>&1 | tail -3
b() { name=$1; shift
  timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-timer "$@" 2>$O/$name.log | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['final_loss'])" 2>/dev/null || { echo "$name FAILED"; tail -5 $O/$name.log; }; }
for w in fpn_attention db res50ppm; do
  for k in 0 1; do b ${w}_ks$k --workload $w --set nt_ksplit=$k; done
done
for k in 0 1; do b crnn_b32_ks$k --workload crnn --batch 32 --set nt_ksplit=$k; done
b crnn_ks1 --workload crnn

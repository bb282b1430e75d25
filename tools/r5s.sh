#!/bin/bash
# split reduction of the 4-wave NT kernels (mr_tuning.nt_ksplit): parity, then in-step A/B
cd /root/repo
mkdir -p gpurun_out/r5s
O=gpurun_out/r5s
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "split_reduction or gemm_nt" 2>&1 | tail -4
b() { name=$1; shift
  timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-kernel-timer "$@" 2>$O/$name.log | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['final_loss'])" 2>/dev/null || { echo "$name FAILED"; tail -5 $O/$name.log; }; }
for w in fpn_attention db; do
  for k in 0 1 4; do b ${w}_ks$k --workload $w --set nt_ksplit=$k; done
done
for k in 0 1; do b crnn_b32_ks$k --workload crnn --batch 32 --set nt_ksplit=$k; b res50ppm_ks$k --workload res50ppm --set nt_ksplit=$k; b crnn_ks$k --workload crnn --set nt_ksplit=$k; done


#!/bin/bash
# PPM head kernels: parity tests, then Res50-PPM / FPN step time
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_res50ppm_gpu.py tests/test_kernels_gpu.py -x -q -k "pool or bilinear or ppm or resize or res50 or cat" 2>&1 | tail -8
for w in res50ppm fpn_attention; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'])"
done

#!/bin/bash
mkdir -p gpurun_out
for b in 0 4 5; do
  echo "== --big $b"
  timeout 200 python tools/microbench_conv.py --layers 6 --only fwd,dgrad --big $b 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/nt_big_variants2.log
echo "== auto, layers 4,5"; timeout 200 python tools/microbench_conv.py --layers 4,5 --only fwd,dgrad 2>&1 | grep -v amdgpu.ids

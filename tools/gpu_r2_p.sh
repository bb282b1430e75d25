#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2p; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_crnn_gpu.py tests/test_ddp_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/tests.log | tee -a $O/summary.txt
for fan in 1 0; do
  MEGREADER_FAN=$fan timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_fan$fan.log 2>&1
  echo "fan=$fan: $(tail -1 $O/bench_fan$fan.log | cut -c1-260)" | tee -a $O/summary.txt
done
for m in 0 1; do echo "== conv wgrad, tnmodel $m" | tee -a $O/summary.txt; timeout 200 python tools/microbench_conv.py --only wgradtab --tnmodel $m 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; done
for sp in 2 3 4 6 8 12 16; do echo "== conv wgrad, splits $sp" | tee -a $O/summary.txt; timeout 200 python tools/microbench_conv.py --only wgradtab --tnsplits $sp 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; done
python tools/microbench_tn_dense.py 2>&1 | grep -v amdgpu.ids | cut -c1-80 | tee -a $O/summary.txt

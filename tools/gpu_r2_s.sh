#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2s; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "big_tile_tn or row_table" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/tests.log | head -20 | tee -a $O/summary.txt
for m in 0 2 1; do echo "== conv wgrad, tnbig $m" | tee -a $O/summary.txt; timeout 200 python tools/microbench_conv.py --only wgradtab --tnbig $m 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; mkdir -p $O
for m in 2 1; do echo "== conv wgrad, tnbig $m, no colsum" | tee -a $O/summary.txt; timeout 200 python tools/microbench_conv.py --only wgradtab --tnbig $m --tnabl 8 --layers 3,5 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; done

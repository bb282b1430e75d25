#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2k; mkdir -p $O
for p8 in 1 2 3 4; do echo "== p8 variant $p8 (1 normal, 2 no-stage, 3 no-reads, 4 mfma only)" | tee -a $O/summary.txt; timeout 300 python tools/microbench_conv.py --only fwd,dgrad --layers 3,5 --p8 $p8 --tnbuf 1 2>&1 | grep -v amdgpu | tee -a $O/summary.txt; done

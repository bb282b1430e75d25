#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
m() { echo "== $*"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --iters 30 "$@" 2>&1 | grep -v "^total\|amdgpu.ids"; }
{
m --layers 1,2,6
for b in 5 8 9 10 11 12 13; do m --layers 1,2,6 --big $b; done
} > $O/conv_sweep2.txt 2>&1
cat $O/conv_sweep2.txt

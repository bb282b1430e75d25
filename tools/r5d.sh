#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
m() { echo "== $*"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --iters 30 "$@" 2>&1 | grep -v "^total\|amdgpu.ids"; }
{
m --layers 2,3,4,5
for b in 8 9 6 7 5; do m --layers 2,3,4,5 --big $b; done
} > $O/conv_sweep3.txt 2>&1
cat $O/conv_sweep3.txt

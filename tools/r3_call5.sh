#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c5; mkdir -p $O
timeout 300 python tools/dbg_block.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tee $O/dbg_block.txt

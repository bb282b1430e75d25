#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c8; mkdir -p $O
L='tests/test_deformable_resnet_gpu.py::test_block_parity'
OMP_NUM_THREADS=32 timeout 200 python -m pytest "$L" -m gpu -q -s 2>&1 | grep -E "^block|passed|failed|AssertionError" | cut -c1-600; echo " <= OMP 32 alone"
timeout 200 python -m pytest "tests/test_dcn_gpu.py::test_real_layer_shapes_vs_oracle" "$L" -m gpu -q -s > $O/combo.log 2>&1; grep -E "^block|passed|failed|AssertionError" $O/combo.log | cut -c1-900
timeout 200 python tools/dbg_block.py 32 2>&1 | grep -v "amdgpu.ids\|Warning\|warn\|rel = " | tee $O/dbg_block.txt

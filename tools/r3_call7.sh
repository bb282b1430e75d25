#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c7; mkdir -p $O
L='tests/test_deformable_resnet_gpu.py::test_block_parity'
for k in test_real_layer_shapes_vs_oracle test_fused_path_equals_general_path test_round2_backward_kernels test_forward_backward_vs_oracle test_extension_level_entry_points_vs_oracle test_dcn_v1_vs_oracle; do
  timeout 200 python -m pytest tests/test_dcn_gpu.py::$k "$L" -m gpu -q -s 2>&1 | grep -E "^block|passed|failed|bad" | tr '\n' ' ' | cut -c1-400; echo " <= $k"
done
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 400 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py -m gpu -q -s 2>&1 | grep -E "^block|passed|failed|bad" | tr '\n' ' ' | cut -c1-600; echo " <= no caching"

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2z; mkdir -p $O
timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/tests.log | head -20 | tee -a $O/summary.txt
timeout 300 python tools/microbench_dcn.py 2>&1 | grep -v amdgpu.ids | tee $O/dcn_microbench.txt | cut -c1-100

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3probe; mkdir -p $O
for v in 0 8 0 8; do
  MEGREADER_DCN_FORK_GFLOP=$v timeout 300 python bench.py --workload db --no-cpu-baseline --no-secondary --no-kernel-timer --steps 40 --warmup 5 > $O/ab_fork$v.log 2>&1
  echo "fork$v db $(tail -1 $O/ab_fork$v.log | grep -o '"ms_per_step": [0-9.]*') $(grep -c Traceback $O/ab_fork$v.log)"
done
MEGREADER_DCN_FORK_GFLOP=8 timeout 600 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log

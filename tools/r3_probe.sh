#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3probe; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_res50ppm_gpu.py tests/test_crnn_gpu.py tests/test_deformable_resnet_gpu.py tests/test_fpn_attention_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
run() { n=$1; shift
  for w in res50ppm db fpn_attention crnn; do
    env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-kernel-timer --steps 40 --warmup 5 > $O/ab_${n}_$w.log 2>&1
    echo "$n $w $(tail -1 $O/ab_${n}_$w.log | grep -o '"ms_per_step": [0-9.]*')"
  done
}
run xmask1 MEGREADER_BN_XMASK=1
run xmask0 MEGREADER_BN_XMASK=0
run xmask1b MEGREADER_BN_XMASK=1

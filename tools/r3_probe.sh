#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3probe; mkdir -p $O
timeout 900 python -m pytest tests/test_dcn_gpu.py tests/test_deformable_resnet_gpu.py tests/test_seg_detector_gpu.py tests/test_published_configs_gpu.py tests/test_ddp_gpu.py -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python bench.py --workload db --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_db.log 2>&1; tail -1 $O/bench_db.log | cut -c1-300
for w in db fpn_attention res50ppm; do
  timeout 300 python tools/trace_glue.py --workload $w > $O/glue_$w.txt 2>&1; grep "top-level" $O/glue_$w.txt
done

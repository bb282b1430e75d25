#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_seg_detector_gpu.py -x -q -m gpu > $O/pytest1.log 2>&1; tail -15 $O/pytest1.log
echo done

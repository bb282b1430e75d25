#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2w; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_res50ppm_gpu.py tests/test_seg_detector_gpu.py -q -x -k "batch_norm or bn or res50 or seg" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; grep -E "^E |passed|failed|Error" $O/tests.log | head -20 | tee -a $O/summary.txt
timeout 400 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; cut -c1-260 $O/bench.json | tee -a $O/summary.txt
python -c "
import json; d=json.load(open('$O/bench.json')); print('secondary', d['secondary']['ms_per_step'], d['secondary']['value'])" | tee -a $O/summary.txt

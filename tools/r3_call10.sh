#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c10; mkdir -p $O
timeout 600 python -m pytest tests/test_dropin_fast_gpu.py tests/test_ctc_decoder_gpu.py tests/test_kernels_gpu.py -m gpu -q -s > $O/pytest.log 2>&1
grep -E "passed|failed|Error|drop-in|CTCDecoder fp32|^E " $O/pytest.log | cut -c1-300 | head -30
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_crnn.log 2>&1; tail -1 $O/bench_crnn.log | cut -c1-250

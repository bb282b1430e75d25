#!/bin/bash
# final record of round 6: full GPU suite, smoke, the default bench line (with the PMC traffic of the committed profiles/)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_final; mkdir -p $O
MEGREADER_TIMED_STEP_DUMP=$PWD/$O/bf16_drift_timed_step.txt timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_full_run.log 2>&1; tail -3 $O/pytest_gpu_full_run.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err; tail -c 300 $O/bench_default_final.json
python bench.py --batch 32 --no-secondary > $O/bench_crnn_b32.json 2> /dev/null

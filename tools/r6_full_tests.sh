#!/bin/bash
# full GPU suite on the final sources (what the driver runs at round end) -> profiles/r06_pytest_gpu_full_run.log
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_full; mkdir -p $O
MEGREADER_TIMED_STEP_DUMP=$PWD/$O/bf16_drift_timed_step.txt timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_full_run.log 2>&1; tail -5 $O/pytest_gpu_full_run.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log

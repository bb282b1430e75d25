#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2i; mkdir -p $O
for ov in 0 1; do
MEGREADER_OVERLAP=$ov timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer > $O/bench_ov$ov.log 2>$O/bench_ov$ov.err
python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r2i/bench_ov$ov.log").read().strip().splitlines()[-1])
    print("overlap=$ov bench ms/step", d["ms_per_step"], "img/s", d["value"], "loss", d["final_loss"])
except Exception as e: print("bench parse failed", e)
PY
done

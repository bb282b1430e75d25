#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.log 2>$O/bench_default.err; echo "default bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r2l/bench_default.log").read().strip().splitlines()[-1])
    print("primary ms/step", d["ms_per_step"], "img/s", d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
    s=d.get("secondary")
    if s: print("secondary", s["metric"], s["ms_per_step"], s["value"], s["roofline"]["kernel"] if s.get("roofline") else None, s["roofline"]["frac"] if s.get("roofline") else None, s.get("step_tflops_per_gpu"))
except Exception as e: print("bench parse failed", e)
PY
tail -3 $O/bench_default.err | cut -c1-300 | tee -a $O/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --workload res50ppm --no-cpu-baseline --steps 10 --warmup 3 > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" 25 > $O/res50_kernel_stats.csv 2>&1; head -40 $O/res50_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt; fi
rm -rf $O/trace

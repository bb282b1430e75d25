#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2j; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_nt or conv" > $O/kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt; tail -4 $O/kernels.log | tee -a $O/summary.txt
for p8 in 0 1; do echo "== p8=$p8" | tee -a $O/summary.txt; timeout 300 python tools/microbench_conv.py --only fwd,dgrad --layers 2,3,4,5 --p8 $p8 --tnbuf 1 2>&1 | grep -v amdgpu | tee -a $O/summary.txt; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>$O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
    d=json.loads(open("gpurun_out/r2j/bench.log").read().strip().splitlines()[-1])
    print("bench ms/step", d["ms_per_step"], "img/s", d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
    for k,v in d["kernels"].items(): print("  ",k,v)
except Exception as e: print("bench parse failed", e)
PY
for mode in capture graph2; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 600 python bench.py --force-ddp --ddp-mode $mode --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timer > $O/bench_ddp_$mode.log 2>$O/bench_ddp_$mode.err; echo "ddp $mode rc=$?" | tee -a $O/summary.txt
tail -1 $O/bench_ddp_$mode.log | cut -c1-600 | tee -a $O/summary.txt; tail -3 $O/bench_ddp_$mode.err | cut -c1-300 | tee -a $O/summary.txt
done
timeout 600 python -m pytest tests/test_ddp_gpu.py -q > $O/ddp.log 2>&1; echo "ddp tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/ddp.log | tee -a $O/summary.txt

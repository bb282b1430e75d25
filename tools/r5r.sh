#!/bin/bash
# is the first bench of a fresh box slower?  (clock ramp / first touch)
cd /root/repo
mkdir -p gpurun_out/r5r
for i in 1 2 3; do python bench.py --no-secondary --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i default', d['ms_per_step'])"; done
python bench.py --no-secondary --no-cpu-baseline --no-kernel-timer --warmup 400 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warmup 400', d['ms_per_step'])"
python bench.py --no-secondary --no-cpu-baseline --no-kernel-timer --steps 400 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 400', d['ms_per_step'])"
rocm-smi --showclocks 2>/dev/null | head -20

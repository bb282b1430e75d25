#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3probe; mkdir -p $O
run() { # name, args...
  n=$1; shift
  for w in crnn res50ppm db fpn_attention; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-secondary --no-kernel-timer --steps 40 --warmup 5 "$@" > $O/ab_${n}_$w.log 2>&1
    echo "$n $w $(tail -1 $O/ab_${n}_$w.log | grep -o '"ms_per_step": [0-9.]*')"
  done
}
run deep1 --set nt_deep=1
run deep0 --set nt_deep=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log

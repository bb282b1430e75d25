#!/bin/bash
# round 6, call 2: parity of the ping-pong 32x32x16 NT kernel + schedule sweep on the CRNN layers
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_2; mkdir -p $O
timeout 900 python -m pytest tests/test_nt32_gpu.py -x -q > $O/test_nt32.log 2>&1; tail -5 $O/test_nt32.log
{
echo "== baseline (nt_m32=0)"; python tools/microbench_conv.py --only fwd,dgrad --layers 1,2,3,4,5 --tune nt_m32=0 2>/dev/null
for shape in 2 3 4 5; do for opt in 10 12 20 21 22 40 42; do
  if [ $shape = 3 ] && [ $opt -lt 20 ]; then continue; fi
  echo "== nt_m32=$shape opt=$opt"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --layers 1,2,3,4,5 --tune nt_m32=$shape,nt_m32_opt=$opt 2>/dev/null
done; done
} > $O/sweep.txt 2>&1
grep -c TF $O/sweep.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_6; mkdir -p $O
timeout 900 python -m pytest tests/test_nt32_gpu.py -x -q > $O/test_nt32.log 2>&1; tail -3 $O/test_nt32.log
python tools/probe_nt_fixed_cost.py "nt_m32=0" "nt_m32=2,nt_m32_opt=20" "nt_m32=2,nt_m32_opt=28" "nt_m32=3,nt_m32_opt=40" "nt_m32=3,nt_m32_opt=48" 2>/dev/null > $O/fixed_cost.txt
cat $O/fixed_cost.txt
{
for t in "nt_m32=0" "nt_m32=2,nt_m32_opt=20" "nt_m32=2,nt_m32_opt=28" "nt_m32=3,nt_m32_opt=40" "nt_m32=3,nt_m32_opt=48" "nt_m32=4,nt_m32_opt=20" "nt_m32=5,nt_m32_opt=20"; do
  echo "== $t"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --layers 1,2,3,4,5 --tune $t 2>/dev/null
done
} > $O/sweep3.txt 2>&1
grep -v amdgpu $O/sweep3.txt

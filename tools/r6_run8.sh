#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_8; mkdir -p $O
python tools/probe_nt_fixed_cost.py "nt_m32=0" "nt_m32=2,nt_m32_opt=20" "nt_m32=2,nt_m32_opt=26" "nt_m32=2,nt_m32_opt=28" 2>/dev/null > $O/fixed_cost.txt
cat $O/fixed_cost.txt
{
for t in "nt_m32=0" "nt_m32=2,nt_m32_opt=20" "nt_m32=2,nt_m32_opt=26" "nt_m32=3,nt_m32_opt=40"; do
  echo "== $t"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --layers 2,3,4,5 --tune $t 2>/dev/null
done
} > $O/sweep4.txt 2>&1
grep -v amdgpu $O/sweep4.txt

#!/bin/bash
# round 6, call 4: wave -> SIMD placement probe; ping-pong kernel with the other group mapping
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_4; mkdir -p $O
tools/wave_simd_map > $O/wave_simd_map.txt 2>&1; cat $O/wave_simd_map.txt
timeout 900 python -m pytest tests/test_nt32_gpu.py -x -q > $O/test_nt32.log 2>&1; tail -3 $O/test_nt32.log
export MEGREADER_HIP_LIB=$PWD/megreader_amd/csrc/libmegreader_hip_abl.so
{
for t in "nt_m32=0" "nt_m32=2,nt_m32_opt=20" "nt_m32=2,nt_m32_opt=24" "nt_m32=2,nt_m32_opt=97" "nt_m32=3,nt_m32_opt=40" "nt_m32=3,nt_m32_opt=44" "nt_m32=3,nt_m32_opt=97" "nt_m32=4,nt_m32_opt=24" "nt_m32=5,nt_m32_opt=24"; do
  echo "== $t"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --layers 1,2,3,4,5 --tune $t 2>/dev/null
done
} > $O/sweep2.txt 2>&1
grep -v amdgpu $O/sweep2.txt

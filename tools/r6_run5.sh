#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_5; mkdir -p $O
export MEGREADER_HIP_LIB=$PWD/megreader_amd/csrc/libmegreader_hip_abl.so
python tools/probe_nt_fixed_cost.py "nt_m32=0" "nt_m32=2,nt_m32_opt=20" "nt_m32=2,nt_m32_opt=94" "nt_m32=2,nt_m32_opt=95" "nt_m32=3,nt_m32_opt=40" "nt_m32=3,nt_m32_opt=94" 2>/dev/null > $O/fixed_cost.txt
cat $O/fixed_cost.txt

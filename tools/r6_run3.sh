#!/bin/bash
# round 6, call 3: nt32 parity (fixed statistics comparison), ablation ladder of the ping-pong kernel, SQ counters of both NT kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_3; mkdir -p $O
timeout 900 python -m pytest tests/test_nt32_gpu.py -x -q > $O/test_nt32.log 2>&1; tail -3 $O/test_nt32.log
export MEGREADER_HIP_LIB=$PWD/megreader_amd/csrc/libmegreader_hip_abl.so
{
for t in "nt_m32=0" "nt_m32=2,nt_m32_opt=0" "nt_m32=2,nt_m32_opt=91" "nt_m32=2,nt_m32_opt=92" "nt_m32=2,nt_m32_opt=93" "nt_m32=3,nt_m32_opt=0" "nt_m32=3,nt_m32_opt=91" "nt_m32=3,nt_m32_opt=92" "nt_m32=3,nt_m32_opt=93"; do
  echo "== $t"; timeout 120 python tools/microbench_conv.py --only fwd,dgrad --layers 3,5 --tune $t 2>/dev/null
done
} > $O/ablation.txt 2>&1
unset MEGREADER_HIP_LIB
for cfg in "nt_m32=0" "nt_m32=1"; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${cfg}_$i -- python tools/pmc_case_nt.py $cfg > $O/pmc_${cfg}_$i.log 2>&1
    f=$(find $O/pmc_${cfg}_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc_sq_${cfg}_pass$i.txt 2>&1; fi
    rm -rf $O/pmc_${cfg}_$i
  done
done
cat $O/ablation.txt | grep -v amdgpu; ls $O

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c4; mkdir -p $O
timeout 300 python tools/dbg_dcn_small.py 2>&1 | grep -v amdgpu.ids | tee $O/dbg_small.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- python tools/microbench_dcn.py --batch 16 --iters 2 --layers layer2.1 > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc_pass$i.txt 2>&1; grep -A9 "dcn2_" $O/pmc_pass$i.txt | head -60; else echo "pmc pass $i failed"; tail -3 $O/pmc$i.log; fi
  rm -rf $O/pmc$i
done

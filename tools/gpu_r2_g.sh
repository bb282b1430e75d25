#!/bin/bash
# kernel trace of the bench + SQ counter passes (MFMA busy, stalls, LDS conflicts, instruction mix)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*" $O/counters_list.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt | tee -a $O/summary.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/trace.log 2>&1
db=$(find $O/trace -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" 47 > $O/kernel_stats.csv 2>&1; head -45 $O/kernel_stats.csv | cut -c1-170 | tee -a $O/summary.txt; fi
rm -rf $O/trace
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- python bench.py --no-cpu-baseline --no-graph --no-kernel-timer --steps 3 --warmup 2 > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc$i.txt 2>&1; else echo "pass $i failed" | tee -a $O/summary.txt; tail -5 $O/pmc$i.log | tee -a $O/summary.txt; fi
  rm -rf $O/pmc$i
done
grep -A9 "igemm_nt_big\|igemm_nt_glds_kernelIDF16bLi96\|igemm_tn_glds_kernelILi2" $O/pmc1.txt | head -60 | tee -a $O/summary.txt

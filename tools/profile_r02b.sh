#!/bin/bash
# Round-2 (second session) measurement on the GPU box: default bench line, rocprofv3 kernel traces of both north-star
# workloads, FETCH_SIZE / WRITE_SIZE PMC passes (separate passes, --pmc never combined with trace domains), two SQ
# counter passes of the CRNN step.  Outputs: gpurun_out/r02b/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02b; mkdir -p $O
timeout 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_default.json; cut -c1-200 $O/bench_default.json
for w in crnn res50ppm; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -- python bench.py --workload $w --no-cpu-baseline --no-secondary > $O/trace_$w.log 2>&1
  db=$(find $O/trace_$w -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $O/${w}_kernel_stats.csv 2>&1; head -4 $O/${w}_kernel_stats.csv | cut -c1-160; fi
  rm -rf $O/trace_$w
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python bench.py --workload $w --no-secondary --no-cpu-baseline --no-graph --no-kernel-timer --steps 3 --warmup 2 > $O/pmc_${w}_$c.log 2>&1
    f=$(find $O/pmc_${w}_$c -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc_${w}_$c.txt 2>&1; fi
    rm -rf $O/pmc_${w}_$c
  done
  python tools/pmc_to_json.py $O/pmc_${w}_FETCH_SIZE.txt $O/pmc_${w}_WRITE_SIZE.txt $O/pmc_traffic_${w}.json > /dev/null 2>&1
done
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $O/sq$i -- python bench.py --no-secondary --no-cpu-baseline --no-graph --no-kernel-timer --steps 3 --warmup 2 > $O/sq$i.log 2>&1
  f=$(find $O/sq$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" > $O/pmc_sq_pass$i.txt 2>&1; else echo "sq pass $i failed"; tail -3 $O/sq$i.log; fi
  rm -rf $O/sq$i
done
grep -A9 "igemm_tn_taps\|igemm_nt_big_kernelIDF16bLi1ELi8" $O/pmc_sq_pass1.txt | head -40
cat $O/pmc_traffic_crnn.json | head -60
for m in capture graph2; do
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 python bench.py --force-ddp --ddp-mode $m --no-cpu-baseline --no-secondary > $O/force_ddp_$m.log 2>&1; tail -1 $O/force_ddp_$m.log > $O/bench_force_ddp_$m.json; cut -c1-160 $O/bench_force_ddp_$m.json
done

#!/bin/bash
# after `gpurun -- bash tools/profile_r06.sh` (+ tools/final_record_r06.sh): copy what should be judged from gpurun_out/ into profiles/
cd "$(dirname "$0")/.."
P=gpurun_out/r06p; F=gpurun_out/r6_final
for f in $P/*_kernel_stats.csv $P/*_step_sequence.txt $P/pmc_*_FETCH_SIZE.txt $P/pmc_*_WRITE_SIZE.txt $P/pmc_traffic_*.json \
         $P/ab_world_gt1_configuration.txt $P/bench_force_ddp_capture.json $P/bench_torchrun_n1.json $P/bench_res50ppm_f32.json \
         $P/mfma_ceiling.txt $P/decode_persist_microbench.txt $P/lstm_phases.txt; do
  [ -f "$f" ] && cp "$f" profiles/r06_$(basename "$f")
done
for f in $F/pytest_gpu_full_run.log $F/bf16_drift_timed_step.txt $F/bench_default_final.json $F/bench_crnn_b32.json; do
  [ -f "$f" ] && cp "$f" profiles/r06_$(basename "$f")
done
ls profiles | grep -c r06_

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2n; mkdir -p $O
for m in 0 1 2 3 4 8 15; do echo "== TN ablation mask $m (1 no-stage 2 no-reads 4 no-atomics 8 no-colsum)" | tee -a $O/summary.txt; timeout 300 python tools/microbench_conv.py --only wgradtab --layers 1,3,5 --tnbuf 1 --tnabl $m 2>&1 | grep -v amdgpu | tee -a $O/summary.txt; done
timeout 900 python bench.py --workload fpn_attention --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_attn.log 2>$O/bench_attn.err; echo "attn bench rc=$?" | tee -a $O/summary.txt; tail -1 $O/bench_attn.log | cut -c1-700 | tee -a $O/summary.txt; tail -4 $O/bench_attn.err | cut -c1-300 | tee -a $O/summary.txt

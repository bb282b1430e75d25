#!/bin/bash
# round 6, call 1: MFMA ceiling + baseline numbers of HEAD on one box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_1; mkdir -p $O
tools/mfma_ceiling > $O/mfma_ceiling.txt 2>&1
python tools/microbench_conv.py --only fwd,dgrad --layers 1,2,3,4,5 > $O/conv_base.txt 2>&1
python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_crnn.json 2>$O/bench_crnn.err
tail -3 $O/mfma_ceiling.txt; cat $O/conv_base.txt | tail -7; tail -c 600 $O/bench_crnn.json

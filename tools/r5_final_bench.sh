#!/bin/bash
# the default bench line (with secondaries and CPU baselines), the 32-crop line, the DCN microbenchmark at batch 16
cd /root/repo
mkdir -p gpurun_out/r05b
O=gpurun_out/r05b
( time python bench.py 2>$O/bench_default.log | tail -1 > $O/bench_default.json ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05b/bench_default.json'))
print('crnn', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])
for s in d.get('secondaries', []):
    print(s['config']['workload'][:40], s['ms_per_step'], s['value'], s['roofline']['kernel'], s['roofline']['frac'], (s.get('cpu_baseline') or {}).get('value'))
PY
python bench.py --batch 32 --no-secondary --no-cpu-baseline 2>$O/bench_crnn_b32.log | tail -1 > $O/bench_crnn_b32.json
python -c "import json; d=json.load(open('$O/bench_crnn_b32.json')); print('crnn b32', d['ms_per_step'], d['value'])"
timeout 400 python tools/microbench_dcn.py --batch 16 2>&1 | grep -v amdgpu.ids > $O/dcn_microbench_b16.txt; tail -2 $O/dcn_microbench_b16.txt

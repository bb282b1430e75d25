#!/bin/bash
# final pass of the round on one box: the whole GPU suite, the profile pass (kernel statistics, step sequences, PMC traffic),
# the default bench line and the 32-crop line
cd /root/repo
mkdir -p gpurun_out/r05f
( time timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/r05f/pytest_gpu_full_run.log 2>&1
tail -6 gpurun_out/r05f/pytest_gpu_full_run.log
bash tools/profile_r05.sh 2>&1 | tail -14
mkdir -p profiles_tmp
O=gpurun_out/r05b
mkdir -p $O
# the PMC json of THIS run is what the bench below should read: stage it where bench.py looks (profiles/), the builder copies it back
cp gpurun_out/r05p/pmc_traffic_crnn.json profiles/r05_pmc_traffic_crnn.json
cp gpurun_out/r05p/pmc_traffic_res50ppm.json profiles/r05_pmc_traffic_res50ppm.json
python bench.py 2>$O/bench_default.log | tail -1 > $O/bench_default.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05b/bench_default.json'))
print('crnn', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])
for s in d.get('secondaries', []):
    print(s['config']['workload'][:40], s['ms_per_step'], s['value'], s['roofline']['kernel'], s['roofline']['frac'], s['roofline']['traffic'])
PY
python bench.py --batch 32 --no-secondary --no-cpu-baseline 2>$O/bench_crnn_b32.log | tail -1 > $O/bench_crnn_b32.json
python -c "import json; d=json.load(open('$O/bench_crnn_b32.json')); print('crnn b32', d['ms_per_step'], d['value'])"

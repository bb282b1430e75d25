#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_16; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_16/bench_default.json').read().strip().splitlines()[-1])
print("crnn", d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline'])
for s in d['secondaries']:
    r=s['roofline']
    print(s['workload'], s['ms_per_step'], s['value'], r['kernel'] if r else None, r['bound'] if r else None, r['frac'] if r else None, (s.get('cpu_baseline') or {}).get('value'))
PY
timeout 300 python bench.py --force-ddp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/bench_force_ddp.json 2> $O/bench_force_ddp.err; tail -c 400 $O/bench_force_ddp.json; tail -2 $O/bench_force_ddp.err
timeout 300 python bench.py --workload res50ppm --dtype f32 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 > $O/bench_res50ppm_f32.json 2> $O/bench_res50ppm_f32.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_res50ppm_f32.json | head -1

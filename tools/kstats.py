"""Print a rocprofv3 kernel-stats csv (tools/profile_r05.sh output) with demangled, shortened kernel names and per-step numbers.
usage: python tools/kstats.py <csv> <steps-in-trace> [top]"""
import csv
import subprocess
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['kernel'].strip('"') != 'TOTAL']
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
names = [r['kernel'].strip('"').replace('.kd', '') for r in rows]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
tot = sum(float(r['total_us']) for r in rows)
print("total %.1f us/step, %d launches/step" % (tot / steps, sum(int(r['calls']) for r in rows) / steps))
for r, d in list(zip(rows, dem))[:top]:
    d = d.replace('mr::', '').replace('__hip_bfloat16', 'bf16').replace('DF16b', 'bf16')
    print('%-92s %6.1f/step %9.1f us/step %8s avg %5s%%' % (d[:92], int(r['calls']) / steps, float(r['total_us']) / steps, r['avg_us'],
                                                           r['percent']))

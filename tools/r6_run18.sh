#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6_18; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $O/tr -- python bench.py --workload fpn_attention --no-graph --no-cpu-baseline --no-secondary --no-kernel-timer --steps 2 --warmup 2 > $O/trace.log 2>&1
python - <<'PY' > gpurun_out/r6_18/memcpy_context.txt 2>&1
import csv, glob, re
O='gpurun_out/r6_18'
api=glob.glob(O+'/tr/*/*hip_api_trace.csv')[0]; ker=glob.glob(O+'/tr/*/*kernel_trace.csv')[0]
kn={}
for r in csv.DictReader(open(ker)):
    kn[r['Correlation_Id']]=re.sub(r'\(.*','',r['Kernel_Name'])[:60]
rows=list(csv.DictReader(open(api)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = after the last adam_kernel launch but one
launch_idx=[i for i,r in enumerate(rows) if r['Function']=='hipLaunchKernel']
adam=[i for i in launch_idx if 'adam_kernel' in kn.get(rows[i]['Correlation_Id'],'')]
start=adam[-2] if len(adam)>=2 else 0
prev='?'
out=[]
for i in range(start, adam[-1]+1):
    r=rows[i]
    f=r['Function']
    if f=='hipLaunchKernel':
        prev=kn.get(r['Correlation_Id'],'?')
    elif 'Memcpy' in f or 'Memset' in f:
        nxt='?'
        for j in range(i+1, min(i+200,len(rows))):
            if rows[j]['Function']=='hipLaunchKernel':
                nxt=kn.get(rows[j]['Correlation_Id'],'?'); break
        out.append((f, prev, nxt))
import collections
print(len(out), "memcpy/memset API calls in the last step")
for k,v in collections.Counter(out).most_common(60): print(v, k)
PY
rm -rf $O/tr; head -70 $O/memcpy_context.txt | cut -c1-220

"""Diagnostic: one training step of a rocprofv3 kernel trace (between two optimizer launches) summed per kernel and per class.
usage: step_breakdown.py <kernel_trace.csv> [rows]"""
import csv, collections, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam' in r['Kernel_Name'].lower()]
ends=[idx[i] for i in range(len(idx)) if i+1==len(idx) or idx[i+1]!=idx[i]+1]
a,b=ends[-3]+1, ends[-2]+1
step=[r for r in rows[a:b]]
def short(n):
    n=re.sub(r'\(.*','',n); n=n.replace('void ','').replace('hf::','')
    return n[:64]
tot=collections.defaultdict(lambda:[0,0])
for r in step:
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    k=short(r['Kernel_Name']); tot[k][0]+=d; tot[k][1]+=1
s=sum(v[0] for v in tot.values())
print("kernels", len(step), "sum kernel ms %.3f"%(s/1e6))
cls=collections.defaultdict(float)
for k,v in tot.items():
    c='gemm(lib)' if k.startswith('Cijk') else 'bn' if k.startswith('bn_') else 'xconv' if 'xconv' in k else 'depthwise' if 'depthwise' in k else 'hf linear/wgrad' if ('linear_' in k or 'wgrad' in k) else 'group/gather' if ('group_point' in k or 'gather' in k) else 'aten' if 'at::' in k else 'other'
    cls[c]+=v[0]/1e6
print({k:round(v,2) for k,v in sorted(cls.items(), key=lambda kv:-kv[1])})
for k,v in sorted(tot.items(), key=lambda kv:-kv[1][0])[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    print("%-66s %4d %8.3f ms"%(k,v[1],v[0]/1e6))

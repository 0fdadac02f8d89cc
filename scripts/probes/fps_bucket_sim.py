"""Diagnostic (CPU, numpy): how many 64-point buckets of the bucketed FPS kernel a round really touches, and how many fall on the
busiest of 16 waves -- the same Morton sort, boxes and skip test as fps_bucket_kernel, 16384 -> 4096."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import kitti_frustum, kitti_uniform
rng=np.random.default_rng(0)
for gen in (kitti_uniform, kitti_frustum):
    pts=gen(rng,1,16384)[0].astype(np.float32)
    n=len(pts); m=4096
    lo=pts.min(0); ext=pts.max(0)-lo
    bits=[0,0,0]; cs=ext.copy()
    for b in range(12):
        d=int(np.argmax(cs)); bits[d]+=1; cs[d]*=0.5
    q=[np.clip(((pts[:,d]-lo[d])*( (1<<bits[d])/ext[d])).astype(int),0,(1<<bits[d])-1) for d in range(3)]
    code=np.zeros(n,int); pos=0
    for b in range(12):
        for d in range(3):
            if b<bits[d]:
                code|=((q[d]>>b)&1)<<pos; pos+=1
    order=np.argsort(code,kind='stable')
    P=pts[order]
    nb=n//64
    B=P.reshape(nb,64,3)
    bmin=B.min(1); bmax=B.max(1)
    td=np.full((nb,64),1e38,np.float32)
    bm=np.full(nb,1e38,np.float32)
    cur=pts[0]
    touched=[]; maxwave=[]
    for j in range(1,m):
        e=np.maximum(np.maximum(bmin-cur,cur-bmax),0)
        lb=(e*e).sum(1)
        need=~(lb*0.99999>bm)
        idx=np.nonzero(need)[0]
        d=((B[idx]-cur)**2).sum(2)
        td[idx]=np.minimum(td[idx],d)
        bm[idx]=td[idx].max(1)
        touched.append(len(idx))
        # wave of bucket g = g % 16
        maxwave.append(np.bincount(idx%16,minlength=16).max() if len(idx) else 0)
        f=np.argmax(td.reshape(-1)); cur=P[f]
    t=np.array(touched); mw=np.array(maxwave)
    print(gen.__name__,"bits",bits,"avg touched",t.mean(),"avg max-per-wave",mw.mean())
    for a,b in ((0,64),(64,256),(256,1024),(1024,2048),(2048,4095)):
        print("  rounds %d-%d: touched %.1f  max/wave %.2f"%(a,b,t[a:b].mean(),mw[a:b].mean()))

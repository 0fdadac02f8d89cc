"""Diagnostic: how fast does torch run the tall-skinny fp32 GEMM / BN shapes of SA level 1?"""
import torch, json
import torch.nn.functional as F
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1)/it*1e3,1)
R=8*4096*32
res={}
for ci,co in ((4,32),(32,32),(32,64),(64,64)):
    x=torch.randn(R,ci,device='cuda'); W=torch.randn(ci,co,device='cuda'); b=torch.randn(co,device='cuda')
    Wt=W.t().contiguous()
    res[f"mm_{ci}x{co}"]=t(lambda: x@W)
    res[f"linear_{ci}x{co}"]=t(lambda: F.linear(x,Wt,b))
    res[f"bmm1024_{ci}x{co}"]=t(lambda: torch.bmm(x.view(1024,-1,ci), W.expand(1024,ci,co)))
    res[f"bmm8192_{ci}x{co}"]=t(lambda: torch.bmm(x.view(8192,-1,ci), W.expand(8192,ci,co)))
    xt=x.t().contiguous()
    res[f"mmT_{ci}x{co}"]=t(lambda: Wt@xt)
    res[f"copy_bytes_{ci}x{co}_MB"]=round((R*ci+R*co)*4/1e6,1)
y=torch.randn(R,32,device='cuda')
bn=torch.nn.BatchNorm1d(32,eps=1e-3).cuda()
res["bn1d_R32"]=t(lambda: bn(y))
res["manual_stats_R32"]=t(lambda: (y.mean(0), y.var(0,unbiased=False)))
res["var_mean_R32"]=t(lambda: torch.var_mean(y,0,unbiased=False))
res["relu_R32"]=t(lambda: torch.relu(y))
res["copy_R32"]=t(lambda: y.clone())
print(json.dumps(res,indent=1))
